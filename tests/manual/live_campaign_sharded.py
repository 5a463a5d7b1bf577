#!/usr/bin/env python3
"""The command line as a job of 2..5 rank processes (tgsfilter --ranks N, emulation build) against the reference binary run side by
side on freshly generated inputs and random flag sets (tests/test_cli_live.py's generators: downsampling, repeat gate, -D, several
adapters, FASTA, .gz output ...): tests/manual/live_campaign_sharded.py <first seed> <last seed> [gpu [reads per case]]
(gpu: the real binary, every rank on device 0)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TGSF_DEBUG_KNOBS"] = "1"
from tests import cli_check
from tests.test_cli_live import case, case2, REF, ROOT
gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
binary = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter") if gpu else os.path.join(ROOT, "tests", "emul", "tgsfilter_emul")
own = ["--devices", "0"] if gpu else []
N = int(sys.argv[4]) if len(sys.argv) > 4 else (300 if gpu else 60)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0; n = 0; both_failed = 0
for seed in range(lo, hi):
    try:
        if seed % 2:
            reads, flags, adapters, fasta = case(seed, N)
            r = cli_check.compare_live(binary, REF, reads, flags, adapters, fasta, ranks=2 + seed % 3, own_args=own)
        else:
            reads, flags, adapters, in_fmt, out_name = case2(seed, N)
            if in_fmt not in ("fq", "fa"):
                in_fmt = "fq"
                out_name = out_name.replace(".fa", ".fq") if "-f" not in flags else out_name
            r = cli_check.compare_live(binary, REF, reads, flags, adapters, in_fmt=in_fmt, out_name=out_name, ranks=2 + seed % 4, own_args=own)
        n += 1
        both_failed += r == "both failed"
    except AssertionError as e:
        bad += 1
        print("SEED", seed, "FAILED:", str(e)[:1500], flush=True)
    except Exception:
        bad += 1
        print("SEED", seed, "ERROR:", traceback.format_exc()[-1500:], flush=True)
print("sharded live campaign (%s): seeds %d..%d, %d compared (%d where both programs refuse), %d differences" % ("GPU, %d reads a case" % N if gpu else "emulation", lo, hi, n, both_failed, bad))
