#!/usr/bin/env python3
"""AddressSanitizer + UBSan and ThreadSanitizer runs of the command line's host side (pipeline, streamed input, mapped
output) over the emulation library, on the goldens with the threaded / streamed / tiny-stride knobs forced.  CPU only
(sanitizers are not available on the GPU pool).  tests/manual/sanitize_host.py"""
import os, subprocess, sys
os.environ.setdefault("TGSF_DEBUG_KNOBS", "1")      # the test settings used below are read only under this switch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cli_check

GOLD = os.path.join(ROOT, "tests", "golden")
THREADED = {"TGSF_BATCH_BYTES": "30000", "TGSF_CTX_PER_DEVICE": "3", "TGSF_FILL_MIN_BYTES": "1", "TGSF_SCAN_BLOCK": "5000",
            "TGSF_STRIDE_BYTES": "70000", "TGSF_SCAN_THREADS": "3", "TGSF_DOWN_MAP_MIN": "1", "TGSF_DOWN_EARLY_MIN": "1"}
STREAMED = {"TGSF_STREAM_MIN_BYTES": "1", "TGSF_CHUNK_BYTES": "20000", "TGSF_FILL_MIN_BYTES": "1", "TGSF_STRIDE_BYTES": "50000",
            "TGSF_CTX_PER_DEVICE": "3"}
CASES = [("ont_zoo", None), ("hifi_zoo", None), ("hifi_bam", None), ("ont_sam", None), ("ont_zoo", "gzip"), ("ont_fasta", "bgzf"),
         ("down_r", None), ("down_gd", None), ("down_F", None), ("fasta_down", None), ("hifi_auto", None), ("ont_auto", None), ("repeat_k21", None), ("huge_adapter", None), ("hifi_bam_auto", None)]
bad = 0
for kind, target, opts in (("asan", "tgsfilter_asan", {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}),
                           ("tsan", "tgsfilter_tsan", {"TSAN_OPTIONS": "halt_on_error=1:report_thread_leaks=0"})):   # the program leaves with _exit: threads are not joined
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), kind], check=True)
    binary = os.path.join(ROOT, "tests", "emul", target)
    for name, comp in CASES:
        for env in (THREADED, STREAMED):
            if env is STREAMED and comp is None and name not in ("hifi_bam", "ont_sam", "hifi_bam_auto"):
                continue
            for k in list(os.environ):
                if k.startswith("TGSF_"):
                    del os.environ[k]
            os.environ["TGSF_DEBUG_KNOBS"] = "1"
            os.environ.update(env)
            os.environ.update(opts)
            try:
                cli_check.run_case(binary, GOLD, name, extra_args=["-t", "8"], compress=comp)
                # ... and as a job of three rank processes (fork, sockets, part files, the pre-pass broadcast, the tally exchange)
                if env is THREADED and comp is None and cli_check.shardable(GOLD, name):
                    cli_check.run_case(binary, GOLD, name, extra_args=["-t", "8"], ranks=3)
                    cli_check.run_case(binary, GOLD, name, extra_args=["-t", "8"], ranks=2, launcher="external")
            except AssertionError as e:
                bad += 1
                print(kind, "FAIL", name, comp, str(e)[-2000:])
    print(kind, "done")
print("failures:", bad)
sys.exit(1 if bad else 0)
