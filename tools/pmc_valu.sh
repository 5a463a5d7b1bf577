#!/bin/bash
# PMC pass: VALU instruction counts and wave-cycle buckets per kernel (own run, kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_valu
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES \
  --output-format csv -d $R/gpurun_out/pmc_valu -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $R/gpurun_out/pmc_valu.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/pmc_valu/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in f:
    for row in csv.DictReader(open(fn)):
        k=row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
        if row["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
with open(R+"/gpurun_out/pmc_valu_summary.txt","w") as o:
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_INSTS_VALU",0)):
        n=max(cnt[k],1)
        o.write(k+" launches=%d "%n+" ".join("%s=%.4g"%(c,x/n) for c,x in sorted(v.items()))+"\n")
PY
