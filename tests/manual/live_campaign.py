#!/usr/bin/env python3
"""One-off campaign: many seeds of tests/test_cli_live.py's generators, the real CLI against the reference binary
(both must be built).  tests/manual/live_campaign.py <first> <last> <reads> [binary]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import cli_check  # noqa: E402
from tests.test_cli_live import REF, case, case2  # noqa: E402

a, b, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
binary = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
bad = both = 0
for seed in range(a, b):
    for gen in (case, case2):
        if gen is case:
            reads, flags, adapters, fasta = gen(seed, n)
            kw = dict(fasta=fasta)
        else:
            reads, flags, adapters, in_fmt, out_name = gen(seed, n)
            kw = dict(in_fmt=in_fmt, out_name=out_name)
        try:
            r = cli_check.compare_live(binary, REF, reads, flags, adapters, **kw)
            both += r == "both failed"
        except AssertionError as e:
            bad += 1
            print(seed, gen.__name__, "FAIL", kw, flags, adapters is not None, str(e)[:600], flush=True)
print("seeds %d..%d: %d failures, %d cases neither binary could finish" % (a, b, bad, both))
