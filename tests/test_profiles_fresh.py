"""profiles/traffic.json (the PMC figures bench.py quotes in roofline.traffic) must come from today's kernel sources: it is stamped
with bench.kernel_source_hash(), and bench.py reports traffic: null / stale_profile: true when the stamp does not match.  This test
makes a change of the kernel sources without a new tools/profile_round.sh sitting visible before the round ends."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_json_is_of_the_current_kernel_sources():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["kernel_source_hash"] == bench.kernel_source_hash(), "re-run tools/round_end_gpu.sh profile (tools/profile_round.sh for the four shapes + tools/merge_traffic.py)"
    # one entry per kernel-path shape bench.py can be asked for, each with the scan's figures
    sigs = t["signatures"]
    assert any(s.startswith("ont:") and ":p=0:" in s for s in sigs) and any(s.startswith("hifi:") for s in sigs)
    for s, d in sigs.items():
        st = d["stages"]["mid_scan"]
        assert st["valu_insts_per_batch"] > 1e8 and st["hbm_bytes_per_batch"] > 1e8, s
