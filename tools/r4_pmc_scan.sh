#!/bin/bash
# round 4: PMC pass over the kernel path (one stream): instruction counts, wave-cycle buckets and the busy-cycle clock of the
# scan kernels.   gpurun -- 'bash tools/r4_pmc_scan.sh <tag> [ENV=VALUE ...]'  -> gpurun_out/r4_pmc_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do export "$v"; done
for pass in ${R4_PASSES:-a b}; do
  rm -rf $R/gpurun_out/r4_pmc_${tag}_$pass
  if [ $pass = a ]; then C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
  elif [ $pass = c ]; then C="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"
  else C="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/r4_pmc_${tag}_$pass -- python3 $R/bench.py --no-e2e --no-cpu-baseline --no-oracle-check --kernel-steps 4 --kernel-warmup 1 --streams 1 > $R/gpurun_out/r4_pmc_${tag}_$pass.log 2>&1
done
python3 - $tag <<'PY'
import csv, glob, collections, os, sys
R=os.environ["GRAFT_REPO_ROOT"]; tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for p in "abc":
    for fn in glob.glob(R+"/gpurun_out/r4_pmc_%s_%s/**/*counter_collection.csv"%(tag,p), recursive=True):
        for row in csv.DictReader(open(fn)):
            k=row["Kernel_Name"].split("(")[0]
            if "k_mid_" in k or "k_stats" in k or "k_end_windows" in k:
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] in ("SQ_WAVES","GRBM_GUI_ACTIVE","SQ_IFETCH"):
                    agg[k]["dur_us_"+row["Counter_Name"][:2]].append((float(row["End_Timestamp"])-float(row["Start_Timestamp"]))/1e3)
with open(R+"/gpurun_out/r4_pmc_%s.txt"%tag,"w") as o:
    for k,v in sorted(agg.items()):
        o.write(k+"\n")
        for c,x in sorted(v.items()):
            x=[y for y in x if y>0.2*max(x)] if max(x)>0 else x
            o.write("   %-24s n=%d avg=%.5g\n"%(c,len(x),sum(x)/max(len(x),1)))
        g=v.get("GRBM_GUI_ACTIVE"); d=v.get("dur_us_GR")
        if g and d:
            gg=[y for y in g if y>0.2*max(g)]; dd=[y for y in d if y>0.2*max(d)]
            o.write("   busy-cycle clock (GRBM_GUI_ACTIVE / 8 XCDs / duration): %.3f GHz\n"%((sum(gg)/len(gg))/8/((sum(dd)/len(dd))*1e3)))
print(open(R+"/gpurun_out/r4_pmc_%s.txt"%tag).read())
PY
rm -rf $R/gpurun_out/r4_pmc_${tag}_a $R/gpurun_out/r4_pmc_${tag}_b $R/gpurun_out/r4_pmc_${tag}_c   # (raw counter files: tens of MB; gpurun_out/ travels back only below 64 MiB)
