#!/usr/bin/env python3
"""Merge the traffic.json of several tools/profile_round.sh sittings (one workload signature each) into profiles/traffic.json.
All of them must carry the hash of the same kernel sources.   python3 tools/merge_traffic.py gpurun_out/prof_r04 gpurun_out/prof_r04_c3 ..."""
import json, os, sys
out = None
for d in sys.argv[1:]:
    t = json.load(open(os.path.join(d, "traffic.json")))
    if out is None:
        out = t
    else:
        assert t["kernel_source_hash"] == out["kernel_source_hash"], (d, t["kernel_source_hash"], out["kernel_source_hash"])
        out["signatures"].update(t["signatures"])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json"), "w"), indent=1)
print(out["kernel_source_hash"], list(out["signatures"]))
