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
    for sink in ("tmpfs_file", "dev_null"):
        s = r["sinks"][sink]
        assert s["same_counters"] and s["gbases_per_s"] > 0 and s["reference_gbases_per_s"] > 0
    assert r["sinks"]["tmpfs_file"]["same_output_multiset"] and r["sinks"]["tmpfs_file"]["output_records"] > 0
    assert r["reads"] == 16 and r["bases"] > 0


def test_write_ont_fastq_is_deterministic(tmp_path):
    from tgsfilter_amd import synth
    a, b = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    ba, na = synth.write_ont_fastq(a, 300, seed=5, mean_len=2000, procs=1, reads_per_job=64)
    bb, nb = synth.write_ont_fastq(b, 300, seed=5, mean_len=2000, procs=3, reads_per_job=64)
    assert (ba, na) == (bb, nb) and open(a, "rb").read() == open(b, "rb").read()
    lines = open(a, "rb").read().split(b"\n")
    assert len(lines) == 4 * 300 + 1 and lines[-1] == b"" and all(l.startswith(b"@r") for l in lines[0:-1:4])
    assert sum(len(l) for l in lines[1:-1:4]) == ba
