#!/bin/bash
# All the rocprofv3 passes behind profiles/: kernel-trace stats for the default bench line and for one batch
# in flight, and the PMC passes (one per counter set, with --kernel-trace only).  Run on the GPU box:
#   tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/*.csv|json  (copy what is to be judged to profiles/)
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# kernel-path runs only under the profiler (the end-to-end leg starts other programs; it is timed by bench.py itself)
K="--no-e2e --no-cpu-baseline --no-oracle-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_default -- python3 $R/bench.py $K > $O/kt_default.json 2> $O/kt_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_single -- python3 $R/bench.py $K --streams 1 > $O/kt_single.json 2> $O/kt_single.err
B="python3 $R/bench.py $K --kernel-steps 2 --kernel-warmup 1 --streams 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- $B > $O/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
python3 - "$O" "$tag" <<'PY'
import csv, glob, collections, json, os, statistics, sys
O, tag = sys.argv[1], sys.argv[2]
def short(n): return n.split("(")[0].replace("void ", "").strip()
def real_dispatches(rows, grid_key):
    """tgsf dispatches in dispatch order, without the miniature warm-up batch of tgsf_prepare_device
    (its k_prepare runs one workgroup; it ends with its k_finalize)."""
    rows = sorted((r for r in rows if "tgsf::" in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
    out, mini, seen = [], False, set()
    for r in rows:
        k = short(r["Kernel_Name"])
        key = (r["Dispatch_Id"])
        if k == "tgsf::k_prepare" and key not in seen:
            mini = int(float(r[grid_key])) <= 256
        seen.add(key)
        if not mini and k != "tgsf::k_noop":
            out.append(r)
        if mini and k == "tgsf::k_finalize":
            mini = False
    return out
# ---- kernel stats (from the per-dispatch trace) ----
for run in ("kt_default", "kt_single"):
    rows = []
    for fn in glob.glob(O + "/" + run + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(fn)))
    by = collections.defaultdict(list)
    for r in real_dispatches(rows, "Grid_Size_X"):
        by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in by.values()) or 1
    with open(O + "/%s_%s_kernel_stats.csv" % (tag, run), "w") as o:
        o.write("Name,Calls,TotalDurationNs,AverageNs,PctOfTgsf,MinNs,MaxNs\n")
        for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            o.write('"%s",%d,%d,%.3f,%.2f,%d,%d\n' % (name, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)))
# ---- PMC ----
def pmc(run):
    rows = []
    for fn in glob.glob(O + "/" + run + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(fn)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for row in real_dispatches(rows, "Grid_Size"):
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
    return {k: {c: v / n[k][c] for c, v in d.items()} for k, d in agg.items()}
fetch, write, tcc, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_tcc"), pmc("pmc_sq")
kernels = sorted(set(fetch) | set(write) | set(tcc), key=lambda k: -tcc.get(k, {}).get("TCC_EA0_RDREQ_sum", 0))
with open(O + "/%s_pmc_hbm_traffic.csv" % tag, "w") as o:
    o.write("kernel,FETCH_SIZE_KB,WRITE_SIZE_KB,TCC_EA0_RDREQ,TCC_HIT,TCC_MISS,read_bytes_corrected,write_bytes\n")
    for k in kernels:
        f = fetch.get(k, {}).get("FETCH_SIZE", 0.0); w = write.get(k, {}).get("WRITE_SIZE", 0.0); t = tcc.get(k, {})
        o.write("%s,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f\n" % (k, f, w, t.get("TCC_EA0_RDREQ_sum", 0), t.get("TCC_HIT_sum", 0),
                                                         t.get("TCC_MISS_sum", 0), t.get("TCC_EA0_RDREQ_sum", 0) * 128, w * 1024))
with open(O + "/%s_pmc_valu.csv" % tag, "w") as o:
    cols = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES"]
    o.write("kernel," + ",".join(cols) + "\n")
    for k, d in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        o.write(k + "," + ",".join("%.0f" % d.get(c, 0) for c in cols) + "\n")
b = json.load(open(O + "/kt_single.json"))
def named(d, prefix):
    """the dispatch-name entry that starts with `prefix` (template arguments vary: k_mid_scan1<2, tgsf::Hot>)"""
    for k in d:
        if k.startswith(prefix):
            return d[k]
    return {}
def hbm(k):
    return named(tcc, k).get("TCC_EA0_RDREQ_sum", 0) * 128 + named(write, k).get("WRITE_SIZE", 0) * 1024
import hashlib
h = hashlib.sha256()
for fn in ("tgsf_hip.h", "tgsf_core.h", "tgsf_dev.h", "tgsf_kernels.h", "tgsf_lib.hip"):      # = bench.py kernel_source_hash()
    h.update(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "tgsfilter_amd", "csrc", fn), "rb").read())
json.dump({"reads_per_step": b["kernel_path"]["reads_per_step"], "kernel_source_hash": h.hexdigest()[:16],
           "mid_scan_hbm_bytes_per_launch": hbm("tgsf::k_mid_scan1<2"),
           "stats_raw_hbm_bytes_per_launch": hbm("tgsf::k_stats<false"),
           "mid_scan_valu_insts_per_launch": named(sq, "tgsf::k_mid_scan1<2").get("SQ_INSTS_VALU"),
           "valu_insts_per_batch_all_kernels": sum(d.get("SQ_INSTS_VALU", 0) for d in sq.values()),
           "source": "profiles/%s_pmc_hbm_traffic.csv (TCC_EA0_RDREQ_sum*128 + WRITE_SIZE*1024) and %s_pmc_valu.csv" % (tag, tag)},
          open(O + "/traffic.json", "w"), indent=1)
PY
ls $O
