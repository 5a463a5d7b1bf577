"""-m gpu: the real command line (tgsfilter_amd/bin/tgsfilter, linked against the HIP library) end to end
against everything the reference produced for the golden cases."""
import os

import pytest

from tests import cli_check, hostmodel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["down_gd", "down_r", "down_F", "fasta_down"])
@pytest.mark.parametrize("how", ["text", "packed"])
def test_cli_gpu_downsampling_qc_pass_both_ways(golden_dir, name, how, monkeypatch):
    """The QC pass over the kept reads reads them in place from the text or packs them first (TGSF_DOWN_QC): same files, same report."""
    monkeypatch.setenv("TGSF_DOWN_QC", how)
    monkeypatch.setenv("TGSF_DOWN_MAP_MIN", "1" if how == "text" else "1000000000000")      # threads into a mapping / the single-stream writer
    cli_check.run_case(os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter"), golden_dir, name, extra_args=["-t", "5"])


@pytest.mark.parametrize("name", hostmodel.GOLDEN_CASES + hostmodel.GOLDEN_CLI_ONLY)
def test_cli_gpu_golden(golden_dir, name):
    binary = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    assert os.path.exists(binary), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    cli_check.run_case(binary, golden_dir, name)


@pytest.mark.parametrize("name", ["ont_zoo", "hifi_auto", "down_r", "ont_fasta"])
def test_cli_gpu_several_contexts(golden_dir, name):
    """--devices 0,0,0: three contexts / feeder threads (on a multi-GPU node these would be different devices)."""
    binary = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    cli_check.run_case(binary, golden_dir, name, extra_args=["--devices", "0,0,0"])


@pytest.mark.parametrize("name", ["ont_zoo", "hifi_zoo", "hifi_bam", "ont_repeat"])
@pytest.mark.parametrize("writer", ["auto", "writev"])
def test_cli_gpu_threaded_pipeline(golden_dir, name, writer, monkeypatch):
    """-t 8, tiny batches, three contexts on the device, several fill threads (or the single-stream writer)."""
    binary = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    monkeypatch.setenv("TGSF_BATCH_BYTES", "30000")
    monkeypatch.setenv("TGSF_CTX_PER_DEVICE", "3")
    monkeypatch.setenv("TGSF_FILL_MIN_BYTES", "1")
    monkeypatch.setenv("TGSF_SCAN_BLOCK", "5000")
    monkeypatch.setenv("TGSF_WRITER", writer)
    monkeypatch.setenv("TGSF_STRIDE_BYTES", "70000")              # the output file grows stride by stride
    cli_check.run_case(binary, golden_dir, name, extra_args=["-t", "8"])


@pytest.mark.parametrize("name,compress", [("ont_zoo", "gzip"), ("hifi_auto", "bgzf"), ("hifi_bam_auto", None), ("ont_sam", None),
                                           ("hifi_fasta_auto", "gzip"), ("repeat_k21", "bgzf")])
def test_cli_gpu_streamed_input(golden_dir, name, compress, monkeypatch):
    """Compressed / BAM / SAM input decoded piece by piece in bounded memory (forced on small files, 20-KB chunks)."""
    binary = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    monkeypatch.setenv("TGSF_STREAM_MIN_BYTES", "1")
    monkeypatch.setenv("TGSF_CHUNK_BYTES", "20000")
    monkeypatch.setenv("TGSF_FILL_MIN_BYTES", "1")
    cli_check.run_case(binary, golden_dir, name, extra_args=["-t", "4"], compress=compress)


REF = os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")
@pytest.mark.parametrize("n_reads,flags", [(800, ["-p", "100", "-k", "11", "-g", "30m", "-d", "2"]), (400, ["-p", "60", "-k", "15"])])
def test_cli_gpu_config5_shape_against_the_reference(tmp_path, n_reads, flags):
    """Config C5's shape at a size the reference finishes in seconds: ultra-long ONT reads (lognormal, mean 150 kb, up to
    2 Mb; 0.06-0.12 Gbases -- the reference's repeat gate runs at 6 Mbases/s per thread), repeat gate and downsampling, the real command line against the reference
    binary (-t 1: its downsampling ties depend on write order) -- byte-equal output, same counters."""
    import subprocess
    import tempfile
    from tgsfilter_amd import synth
    binary = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    shm = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=shm) as td:
        fq = os.path.join(td, "c5.fq")
        bases, _ = synth.write_ont_fastq(fq, n_reads, seed=5, mean_len=150000.0, max_len=2_000_000, reads_per_job=64)
        assert bases > 4e7
        fa = os.path.join(td, "rapid.fa")
        open(fa, "wb").write(b">rapid\n" + synth.ONT_RAPID + b"\n")
        common = ["-i", fq, "-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0", "-a", fa] + flags
        outs = {}
        for tag, exe, t in (("ref", REF, "1"), ("ours", binary, "16")):
            out = os.path.join(td, tag + ".fq")
            p = subprocess.run([exe, "-o", out, "-t", t] + common, capture_output=True)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            info = [l for l in p.stderr.decode().splitlines() if l.startswith("INFO: ") and "written to" not in l and "input adapter" not in l]
            outs[tag] = (open(out, "rb").read(), info)
        assert outs["ours"][1] == outs["ref"][1]
        assert outs["ours"][0] == outs["ref"][0]
        assert len(outs["ref"][0]) > 1e6


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")
def test_cli_gpu_three_feeder_sets_on_a_real_file(tmp_path):
    """What a multi-GPU run does on the host, on the one device of this box: --devices 0,0,0 = three feeder sets (nine
    contexts) pulling batches of a 24 000-read file (C2's shape at 6 kb: 0.14 Gbases, ~290 MB of text, hundreds of batches
    at the forced batch size) in whatever order they finish, the planner putting the records back into input order, the
    tallies of all contexts merged on the host.  Output file, INFO lines and report must equal the --devices 0 run's byte
    for byte, and the reference binary's (-t 1: input order)."""
    import hashlib
    import subprocess
    import tempfile
    from tgsfilter_amd import synth
    binary = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    shm = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=shm) as td:
        fq = os.path.join(td, "in.fq")
        bases, _ = synth.write_ont_fastq(fq, 24_000, seed=12, mean_len=6000.0, max_len=200_000, reads_per_job=512)
        assert bases > 1e8
        common = ["-i", fq, "-x", "ont", "-l", "1000", "-q", "10"]               # automatic pre-pass, as C2
        env = dict(os.environ, TGSF_BATCH_BYTES="1500000", TGSF_EARLY_OPEN_MIN="1", TGSF_STRIDE_BYTES="8000000")

        def run(tag, exe, extra, e=None):
            out = os.path.join(td, tag + ".fq")
            p = subprocess.run([exe, "-o", out] + common + extra, capture_output=True, env=e)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            info = [l for l in p.stderr.decode().splitlines() if l.startswith("INFO: ") and "written to" not in l]
            html = open(os.path.join(td, tag + ".html"), "rb").read()
            # the report names its input and carries a time stamp in its footer: compare the tables and the plotted data
            body = b"\n".join(l for l in html.splitlines() if b"<tr>" in l or l.lstrip().startswith(b"var data"))
            return hashlib.sha256(open(out, "rb").read()).hexdigest(), info, hashlib.sha256(body).hexdigest()

        one = run("one", binary, ["-t", "16", "--devices", "0"], env)
        three = run("three", binary, ["-t", "16", "--devices", "0,0,0"], env)
        ref = run("ref", REF, ["-t", "1"])
        assert three[1] == one[1] == ref[1]
        assert three[0] == one[0] == ref[0]
        assert three[2] == one[2] == ref[2]
