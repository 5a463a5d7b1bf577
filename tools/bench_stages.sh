#!/bin/bash
# usage: tools/bench_stages.sh <logname> [bench args...]   -- runs GPU tests + bench and prints the stage table
log=$1; shift
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gputests.log
python bench.py "$@" > gpurun_out/$log 2>&1
python - "$log" <<PY
import json,sys
l=[x for x in open("gpurun_out/"+sys.argv[1]) if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("value %.1f Gbases/s  ms/step %.3f  dom frac %.3f pipeline frac %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["pipeline_frac"]))
    for k,v in d["roofline"]["stage_ms_per_step"].items(): print("  %-32s %.3f"%(k,v))
    if "cpu_baseline" in d: print("  cpu:",d["cpu_baseline"])
else: print(open("gpurun_out/"+sys.argv[1]).read()[-3000:])
PY
