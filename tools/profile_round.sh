#!/bin/bash
# All the rocprofv3 passes behind profiles/: kernel-trace stats for the default bench line and for one batch
# in flight, and the PMC passes (one per counter set, with --kernel-trace only).  Run on the GPU box:
#   tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/*.csv|json  (copy what is to be judged to profiles/)
#   tools/profile_round.sh <tag> [bench.py arguments of the kernel path: --workload hifi | --min-repeat 100 --kmer 11 ...]
tag=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# kernel-path runs only under the profiler (the end-to-end leg starts other programs; it is timed by bench.py itself)
K="--no-e2e --no-cpu-baseline --no-oracle-check $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_default -- python3 $R/bench.py $K --detail-file $O/kt_default_detail.json > $O/kt_default.json 2> $O/kt_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_single -- python3 $R/bench.py $K --streams 1 --detail-file $O/kt_single_detail.json > $O/kt_single.json 2> $O/kt_single.err
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
def per_batch(run):
    """per kernel name: the counters summed over a batch's dispatches of it (a batch = one k_finalize)"""
    rows = []
    for fn in glob.glob(O + "/" + run + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(fn)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nb = collections.Counter()
    for row in real_dispatches(rows, "Grid_Size"):
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if k == "tgsf::k_finalize": nb[row["Counter_Name"]] += 1
    return {k: {c: v / max(nb[c], 1) for c, v in d.items()} for k, d in agg.items()}
fetch, write, tcc, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_tcc"), pmc("pmc_sq")
b_write, b_tcc, b_sq = per_batch("pmc_write"), per_batch("pmc_tcc"), per_batch("pmc_sq")
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
b = json.load(open(O + "/kt_single_detail.json"))      # (the full record: stdout carries the small line only)
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
STAGES = {"mid_scan": ("tgsf::k_mid_flat", "tgsf::k_mid_scan", "tgsf::k_mid_recheck", "tgsf::k_mid_marks"), "stats_raw": ("tgsf::k_stats<false",), "stats_clean": ("tgsf::k_stats<true",),
          "repeat_gate": ("tgsf::k_repeat",), "end_windows": ("tgsf::k_end_windows",)}
def stage_sum(d, prefixes, counter, scale=1.0):
    return sum(v.get(counter, 0.0) * scale for k, v in d.items() if k.startswith(prefixes))
per_stage = {}
for st, pre in STAGES.items():
    rd, wr = stage_sum(b_tcc, pre, "TCC_EA0_RDREQ_sum", 128.0), stage_sum(b_write, pre, "WRITE_SIZE", 1024.0)
    vi = stage_sum(b_sq, pre, "SQ_INSTS_VALU")
    if rd or wr or vi:
        per_stage[st] = {"hbm_bytes_per_batch": rd + wr, "hbm_read_bytes_per_batch": rd, "valu_insts_per_batch": vi}
json.dump({"kernel_source_hash": h.hexdigest()[:16],
           "signatures": {b["kernel_path"]["signature"]: {
               "reads_per_step": b["kernel_path"]["reads_per_step"], "stages": per_stage,
               "hbm_bytes_per_batch_all_kernels": sum(v.get("TCC_EA0_RDREQ_sum", 0) * 128 for v in b_tcc.values()) + sum(v.get("WRITE_SIZE", 0) * 1024 for v in b_write.values()),
               "valu_insts_per_batch_all_kernels": sum(d.get("SQ_INSTS_VALU", 0) for d in b_sq.values()),
               "source": "profiles/%s_pmc_hbm_traffic.csv (TCC_EA0_RDREQ_sum*128 + WRITE_SIZE*1024) and %s_pmc_valu.csv" % (tag, tag)}}},
          open(O + "/traffic.json", "w"), indent=1)
PY
rm -rf $O/kt_default $O/kt_single $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_sq      # (raw traces: tens of MB; gpurun_out/ travels back only below 64 MiB)
ls $O
