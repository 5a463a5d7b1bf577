#!/usr/bin/env python3
"""The command line as a job of 2..5 rank processes (tgsfilter --ranks N, emulation build) against the reference binary run side by
side on freshly generated inputs and random flag sets (tests/test_cli_live.py's generators: downsampling, repeat gate, -D, several
adapters, FASTA, .gz output ...): tests/manual/live_campaign_sharded.py <first seed> <last seed>."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TGSF_DEBUG_KNOBS"] = "1"
from tests import cli_check
from tests.test_cli_live import case, case2, REF, ROOT
binary = os.path.join(ROOT, "tests", "emul", "tgsfilter_emul")
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0; n = 0; both_failed = 0
for seed in range(lo, hi):
    try:
        if seed % 2:
            reads, flags, adapters, fasta = case(seed, 60)
            r = cli_check.compare_live(binary, REF, reads, flags, adapters, fasta, ranks=2 + seed % 3)
        else:
            reads, flags, adapters, in_fmt, out_name = case2(seed, 60)
            if in_fmt not in ("fq", "fa"):
                in_fmt = "fq"
                out_name = out_name.replace(".fa", ".fq") if "-f" not in flags else out_name
            r = cli_check.compare_live(binary, REF, reads, flags, adapters, in_fmt=in_fmt, out_name=out_name, ranks=2 + seed % 4)
        n += 1
        both_failed += r == "both failed"
    except AssertionError as e:
        bad += 1
        print("SEED", seed, "FAILED:", str(e)[:1500], flush=True)
    except Exception:
        bad += 1
        print("SEED", seed, "ERROR:", traceback.format_exc()[-1500:], flush=True)
print("sharded live campaign: seeds %d..%d, %d compared (%d where both programs refuse), %d differences" % (lo, hi, n, both_failed, bad))
