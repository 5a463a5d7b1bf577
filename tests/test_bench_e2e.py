"""bench.py's end-to-end leg on a GPU-less box: the same code (file generation, both sinks, the reference on the same
file, multiset and counter comparison) with the emulation build of the command line standing in for bin/tgsfilter."""
import os
import subprocess
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")
def test_e2e_leg_with_the_emulated_cli(monkeypatch):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), "emul"], check=True)
    import bench
    monkeypatch.setattr(bench, "CLI", os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"))
    args = types.SimpleNamespace(e2e_reads=16, steps=1, warmup=0, no_cpu_baseline=False, pinned_variant=True)
    r = bench.e2e_leg(args, 1)
    s = r["sinks"]["tmpfs_file"]
    assert s["same_counters"] and s["gbases_per_s"] > 0 and s["reference_gbases_per_s"] > 0
    assert s["same_output_multiset"] and s["output_records"] > 0 and s["files"] == 1 and len(s["per_file"]) == 1
    assert any("5' adapter: GTTTTCGC" in l for l in s["info_prepass"])          # the automatic pre-pass ran (configs[1] as written)
    d = r["sinks"]["dev_null"]
    assert d["same_counters_as_the_file_run"] and d["gbases_per_s"] > 0
    # the program path of N GPUs on one: three rank processes, a part file each, same records and INFO lines
    sh = r["sinks"]["tmpfs_part_files_3_ranks_one_gpu"]
    assert sh["ranks"] == 3 and sh["same_counters_as_the_file_run"] and sh["same_output_multiset"] and len(sh["shard_lines"]) == 3
    v = r["variants"]["pinned_prepass"]
    assert v["same_counters"] and v["same_output_multiset"] and "-5 0 -3 0 -a rapid.fa" in v["flags"]
    assert r["reads"] == 16 and r["bases"] > 0 and r["flags"] == "-x ont -l 1000 -q 10" and not r["skipped"]
    # the reference is timed at its own thread clamp and at what the box's CPU quota runs unthrottled (when they differ);
    # the baseline is the faster of the two
    rbt = s["reference_runs_by_threads_first_file"]
    assert 1 <= len(rbt) <= 2 and all(len(v["wall_s_runs"]) == bench.REF_RUNS for v in rbt.values())
    assert s["reference_threads"] == max(rbt.values(), key=lambda v: v["gbases_per_s"])["threads"]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")
def test_e2e_leg_over_several_files_and_as_rank_processes(monkeypatch):
    """C2 as written does not fit the box at once: its reads are staged as consecutive files, a step is one pass over all
    of them.  And N > 1: the command line runs as N rank processes with a part file each (here: on the emulation)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), "emul"], check=True)
    import bench
    monkeypatch.setattr(bench, "CLI", os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"))
    args = types.SimpleNamespace(e2e_reads=25, e2e_files=3, steps=4, warmup=1, no_cpu_baseline=False, sharded_leg=False)
    r = bench.e2e_leg(args, 1)
    s = r["sinks"]["tmpfs_file"]
    assert r["files"] == 3 and r["reads_per_file"] == [9, 8, 8] and r["reads"] == 25 and len(s["per_file"]) == 3
    # a step = one run over one file; the K = 4 timed steps are dealt over the 3 files in staging order
    assert r["steps_per_file"] == [2, 1, 1] and [len(f["wall_s"]) for f in s["per_file"]] == [2, 1, 1] and s["runs"] == 4
    assert s["wall_s"] == [w for f in s["per_file"] for w in f["wall_s"]]
    filtered = sum(f["bases"] * len(f["wall_s"]) for f in s["per_file"])
    assert abs(s["gbases_per_s"] - filtered / sum(s["wall_s"]) / 1e9) < 1e-12 and abs(s["bases_per_step_mean"] - filtered / 4) < 1e-6
    assert r["bases"] == sum(f["bases"] for f in s["per_file"]) and s["same_counters"] and s["same_output_multiset"]
    assert set(r["seconds_by_phase"]) == {"generate", "ours", "reference", "remove_outputs", "digests"}
    assert abs(s["reference_wall_s"] - sum(f["reference_wall_s"] for f in s["per_file"])) < 1e-9 and s["speedup_vs_reference"] > 0
    assert "tmpfs_part_files_3_ranks_one_gpu" not in r["sinks"]
    # fewer steps than files: only as many files are staged as there are steps
    args = types.SimpleNamespace(e2e_reads=25, e2e_files=3, steps=2, warmup=0, no_cpu_baseline=False, sharded_leg=False)
    r = bench.e2e_leg(args, 1)
    assert r["files"] == 2 and r["files_of_config_split"] == 3 and r["reads"] == 17 and r["steps_per_file"] == [1, 1]
    args = types.SimpleNamespace(e2e_reads=20, e2e_ranks=2, steps=1, warmup=0, no_cpu_baseline=False)
    r = bench.e2e_leg(args, 1)
    s = r["sinks"]["tmpfs_file"]
    assert r["ranks"] == 2 and s["ranks"] == 2 and s["same_counters"] and s["same_output_multiset"] and len(s["per_file"][0]["shard_lines"]) == 2
    assert r["sinks"]["dev_null"]["same_counters_as_the_file_run"]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")
def test_e2e_leg_config_c5_with_the_emulated_cli(monkeypatch):
    """--config c5: ultra-long reads, -g 3g -d 40 -p 100 -k 11 as BASELINE.json writes it, the reference beside it."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), "emul"], check=True)
    import bench
    monkeypatch.setattr(bench, "CLI", os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"))
    args = types.SimpleNamespace(e2e_reads=10, steps=1, warmup=0, no_cpu_baseline=False, config="c5")
    r = bench.e2e_leg(args, 1)
    s = r["sinks"]["tmpfs_file"]
    assert r["flags"] == "-x ont -l 1000 -q 10 -g 3g -d 40 -p 100 -k 11" and "C5" in r["config"]
    assert s["same_counters"] and s["same_output_multiset"] and s["output_records"] > 0 and "variants" not in r


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")
def test_e2e_leg_config_c3_with_the_emulated_cli(monkeypatch):
    """--config c3: HiFi reads, -x hifi -l 1000 -q 20 -M 35 -T 50 with the automatic pre-pass, the reference beside it."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), "emul"], check=True)
    import bench
    monkeypatch.setattr(bench, "CLI", os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"))
    args = types.SimpleNamespace(e2e_reads=40, steps=1, warmup=0, no_cpu_baseline=False, config="c3")
    r = bench.e2e_leg(args, 1)
    s = r["sinks"]["tmpfs_file"]
    assert r["flags"] == "-x hifi -l 1000 -q 20 -M 35 -T 50" and "C3" in r["config"]
    assert s["same_counters"] and s["same_output_multiset"] and s["output_records"] > 0 and "variants" not in r


def test_the_stdout_line_is_small_and_machine_readable(tmp_path):
    """BENCH_r05.json: parsed = null -- round 5's line had grown to 21 kB.  The line carries the contract's keys, `roofline`,
    `cpu_baseline`, the speed-up and the scaling figures in at most 4 kB; the whole record goes to the detail file."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_record_r05.json")))       # the 21-kB record of round 5
    # worst case on top: sentence-long strings everywhere
    full["config"]["workload"] = full["config"]["workload"] * 3
    full["cpu_baseline"]["sample"] = full["cpu_baseline"]["sample"] * 3
    full["bench_wall_s"] = 1234.5
    detail = str(tmp_path / "sub" / "detail.json")
    line = bench.emit(full, detail)
    assert "\n" not in line and len(line.encode()) <= bench.LINE_LIMIT < 6000
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in c
    assert c["value"] == pytest.approx(full["value"], rel=1e-5) and c["vs_baseline"] is None and c["dtype"] == "u8"
    assert len(c["config"]["workload"]) <= 300 and "model" not in c["config"]
    r = c["roofline"]
    assert r["bound"] in ("hbm", "valu", "mfma", "latency") and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert {"kernel", "stage", "kernel_ms", "algorithmic_bytes_per_launch", "traffic", "traffic_all_kernels_per_batch", "fractions_of_hbm_peak"} <= set(r)
    b = c["cpu_baseline"]
    assert b["kind"] == "reference" and b["cores"] >= 1 and b["value"] > 0 and len(b["sample"]) <= 200 and set(b) == {"value", "unit", "cores", "kind", "sample"}
    assert c["e2e_speedup_vs_reference"]["tmpfs_file"] == pytest.approx(full["e2e_speedup_vs_reference"]["tmpfs_file"], rel=1e-5)
    assert c["scaling_figures"]["n_gpus"] == 1 and c["detail"] == detail
    assert json.load(open(detail))["e2e"]["sinks"]["tmpfs_file"]["per_file"][0]["wall_s"] == full["e2e"]["sinks"]["tmpfs_file"]["per_file"][0]["wall_s"]


def test_write_ont_fastq_is_deterministic(tmp_path):
    from tgsfilter_amd import synth
    a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    ba, na = synth.write_ont_fastq(a, 300, seed=5, mean_len=2000, procs=1, reads_per_job=64)
    bb, nb = synth.write_ont_fastq(b, 300, seed=5, mean_len=2000, procs=3, reads_per_job=64)
    assert (ba, na) == (bb, nb) and open(a, "rb").read() == open(b, "rb").read()
    lines = open(a, "rb").read().split(b"\n")
    assert len(lines) == 4 * 300 + 1 and lines[-1] == b"" and all(l.startswith(b"@r") for l in lines[0:-1:4])
    assert sum(len(l) for l in lines[1:-1:4]) == ba
