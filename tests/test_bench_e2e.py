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
    args = types.SimpleNamespace(e2e_reads=16, steps=1, warmup=0, no_cpu_baseline=False)
    r = bench.e2e_leg(args, 1)
    s = r["sinks"]["tmpfs_file"]
    assert s["same_counters"] and s["gbases_per_s"] > 0 and s["reference_gbases_per_s"] > 0
    assert s["same_output_multiset"] and s["output_records"] > 0
    assert s["exit_mode"] == "sync" and len(s["detached_wall_s"]) == 1 and s["speedup_vs_reference_detached"] > 0
    assert any("5' adapter: GTTTTCGC" in l for l in s["info_prepass"])          # the automatic pre-pass ran (configs[1] as written)
    d = r["sinks"]["dev_null"]
    assert d["same_counters_as_the_file_run"] and d["gbases_per_s"] > 0
    v = r["variants"]["pinned_prepass"]
    assert v["same_counters"] and v["same_output_multiset"] and "-5 0 -3 0 -a rapid.fa" in v["flags"]
    assert r["reads"] == 16 and r["bases"] > 0 and r["flags"] == "-x ont -l 1000 -q 10" and not r["skipped"]
    # the reference is timed several times (mean and best both reported, the cgroup's CPU accounting of every run)
    assert len(s["reference_wall_s_runs"]) == bench.REF_RUNS == len(s["reference_cpu_runs"])
    assert s["reference_wall_s_min"] <= s["reference_wall_s"] and s["speedup_vs_reference_best_run"] <= s["speedup_vs_reference"] * 1.0000001


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


def test_write_ont_fastq_is_deterministic(tmp_path):
    from tgsfilter_amd import synth
    a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    ba, na = synth.write_ont_fastq(a, 300, seed=5, mean_len=2000, procs=1, reads_per_job=64)
    bb, nb = synth.write_ont_fastq(b, 300, seed=5, mean_len=2000, procs=3, reads_per_job=64)
    assert (ba, na) == (bb, nb) and open(a, "rb").read() == open(b, "rb").read()
    lines = open(a, "rb").read().split(b"\n")
    assert len(lines) == 4 * 300 + 1 and lines[-1] == b"" and all(l.startswith(b"@r") for l in lines[0:-1:4])
    assert sum(len(l) for l in lines[1:-1:4]) == ba
