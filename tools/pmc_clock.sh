#!/bin/bash
# effective clock and stall buckets of the scan: GRBM_GUI_ACTIVE (cycles the GPU was busy) per launch / duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_clock
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_clock -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $R/gpurun_out/pmc_clock.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(R+"/gpurun_out/pmc_clock/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        k=row["Kernel_Name"].split("(")[0]
        if "tgsf::k_mid_scan1" in k or "tgsf::k_stats" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            agg[k]["dur_ns"].append(float(row["End_Timestamp"])-float(row["Start_Timestamp"]))
with open(R+"/gpurun_out/pmc_clock.txt","w") as o:
    for k,v in agg.items():
        o.write(k+"\n")
        for c,x in v.items():
            x=[y for y in x if y>0.2*max(x)]
            o.write("   %-20s n=%d avg=%.4g\n"%(c,len(x),sum(x)/len(x)))
PY
