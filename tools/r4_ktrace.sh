#!/bin/bash
# round 4: kernel-trace durations of the kernel path (one stream).  gpurun -- 'bash tools/r4_ktrace.sh <tag> [ENV=VALUE ...]' -> gpurun_out/r4_kt_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do export "$v"; done
rm -rf $R/gpurun_out/r4_kt_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_kt_$tag -- python3 $R/bench.py --no-e2e --no-cpu-baseline --no-oracle-check --kernel-steps ${R4_STEPS:-12} --kernel-warmup 2 --streams ${R4_STREAMS:-1} ${R4_ARGS} > $R/gpurun_out/r4_kt_$tag.log 2>&1
python3 - $tag <<'PY'
import csv, glob, os, sys, collections
R=os.environ["GRAFT_REPO_ROOT"]; tag=sys.argv[1]
d=collections.defaultdict(list)
for fn in glob.glob(R+"/gpurun_out/r4_kt_%s/**/*kernel_trace.csv"%tag, recursive=True):
    for row in csv.DictReader(open(fn)):
        d[row["Kernel_Name"].split("(")[0]].append((float(row["End_Timestamp"])-float(row["Start_Timestamp"]))/1e3)
with open(R+"/gpurun_out/r4_kt_%s.txt"%tag,"w") as o:
    tot=0
    for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
        v2=sorted(v)[len(v)//8: len(v)-len(v)//8] or v          # (drop warm-up outliers)
        o.write("%-70s n=%4d avg_us=%9.1f  min=%9.1f\n"%(k[:70],len(v),sum(v2)/len(v2),min(v)))
print(open(R+"/gpurun_out/r4_kt_%s.txt"%tag).read())
PY
rm -rf $R/gpurun_out/r4_kt_$tag
