#!/usr/bin/env python3
"""One-off campaign: many seeds of tests/fuzz.py (random parameter sets, the library against the oracle, every record and
tally word).  tests/manual/fuzz_campaign.py <first> <last> <reads> [emul]   (TGSF_FUZZ_GATE_P=0.6: repeat gate in 60 % of the cases)"""
import os
os.environ.setdefault("TGSF_DEBUG_KNOBS", "1")      # the test settings used below are read only under this switch
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import fuzz  # noqa: E402

a, b, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = os.path.join(ROOT, "tests", "emul", "libtgsf_emul.so") if len(sys.argv) > 4 and sys.argv[4] == "emul" else None
bad = gated = 0
for seed in range(a, b):
    gated += "min_repeat" in fuzz.random_case(seed, 4)[2]
    try:
        fuzz.run_case(lib, seed, n)
    except AssertionError as e:
        bad += 1
        print(str(e)[:900], flush=True)
    except Exception as e:                                  # the library refused the batch: say which case
        bad += 1
        print("seed %d: %s %s" % (seed, type(e).__name__, str(e)[:300]), {k: v for k, v in fuzz.random_case(seed, 4)[2].items() if k != "adapters"}, flush=True)
print("seeds %d..%d (%d reads each, %d of them with the repeat gate): %d failures" % (a, b, n, gated, bad))
