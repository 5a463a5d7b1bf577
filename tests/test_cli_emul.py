"""The command line end to end on a GPU-less box: tests/emul/tgsfilter_emul is the host program of
tgsfilter_amd/host linked against the serial emulation of the kernels (test infrastructure).  Checked
against everything the reference produced for the golden cases, including the pre-pass (auto trims,
adapter identification through the library's alignment entry point) and the HTML data."""
import os
import subprocess

import pytest

from tests import cli_check, hostmodel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def binary():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), "emul"], check=True)
    return os.path.join(ROOT, "tests", "emul", "tgsfilter_emul")


@pytest.mark.parametrize("name", hostmodel.GOLDEN_CASES + hostmodel.GOLDEN_CLI_ONLY)
def test_cli_golden(binary, golden_dir, name):
    cli_check.run_case(binary, golden_dir, name)


@pytest.mark.parametrize("name", ["down_gd", "down_r", "down_R", "down_F", "fasta_down"])
@pytest.mark.parametrize("how", ["text", "packed"])
def test_cli_downsampling_qc_pass_both_ways(binary, golden_dir, name, how, monkeypatch):
    """The QC pass over the kept reads reads them in place from the text or packs them first (TGSF_DOWN_QC): same files, same report.
    The records are written by threads into a mapping of the file (forced here: TGSF_DOWN_MAP_MIN) or by the single-stream writer."""
    monkeypatch.setenv("TGSF_DOWN_QC", how)
    monkeypatch.setenv("TGSF_DOWN_MAP_MIN", "1" if how == "text" else "1000000000000")
    cli_check.run_case(binary, golden_dir, name, extra_args=["-t", "5"])


@pytest.mark.parametrize("name", ["down_gd", "down_r", "down_R", "down_F", "fasta_down"])
def test_cli_downsampling_output_reserved_early(binary, golden_dir, name, monkeypatch):
    """The downsampling output is instantiated while the filter pass still runs (forced here on the small goldens: tiny strides,
    speculation beyond what the selection keeps -- the surplus is cut off) and filled stride by stride."""
    monkeypatch.setenv("TGSF_DOWN_EARLY_MIN", "1")
    monkeypatch.setenv("TGSF_STRIDE_BYTES", "40000")
    monkeypatch.setenv("TGSF_DOWN_FEEDERS", "2")          # ... and the QC pass over the kept reads from two feeders (small slices: several batches)
    monkeypatch.setenv("TGSF_DOWN_QC", "text")
    monkeypatch.setenv("TGSF_DOWN_BATCH_BYTES", "1200000")          # (a slice ends 1 MB short of it: 150 KB of text each)
    cli_check.run_case(binary, golden_dir, name, extra_args=["-t", "5"])


@pytest.mark.parametrize("name", ["ont_zoo", "hifi_zoo", "ont_auto", "hifi_phred64_auto", "ont_e1300"])
def test_cli_output_created_before_the_prepass(binary, golden_dir, name, monkeypatch):
    """A large run creates its output (when there is none yet) and starts instantiating its pages beside the pre-pass; forced here
    on the goldens, with tiny strides: same files."""
    monkeypatch.setenv("TGSF_EARLY_OPEN_MIN", "1")
    monkeypatch.setenv("TGSF_STRIDE_BYTES", "50000")
    cli_check.run_case(binary, golden_dir, name, extra_args=["-t", "4"])


def test_cli_output_created_early_is_removed_when_the_prepass_ends_the_run(binary, golden_dir, tmp_path, monkeypatch):
    """-q at or above the largest quality of the sample ends the run in the pre-pass (Get_qType, src/TGSFilter.cpp:1063-1068): the reference
    has not opened its output by then.  An output this program created ahead of the pre-pass is removed again; an existing file is not touched."""
    import gzip
    monkeypatch.setenv("TGSF_EARLY_OPEN_MIN", "1")
    fin = tmp_path / "in.fq"
    fin.write_bytes(gzip.open(os.path.join(golden_dir, "ont_zoo.in.fq.gz"), "rb").read())
    out = tmp_path / "out.fq"
    p = subprocess.run([binary, "-i", str(fin), "-o", str(out), "-x", "ont", "-q", "90"], capture_output=True, timeout=120)
    assert p.returncode == 255 and b"Please reset -q parameter" in p.stderr and not out.exists()
    out.write_bytes(b"precious\n")
    p = subprocess.run([binary, "-i", str(fin), "-o", str(out), "-x", "ont", "-q", "90"], capture_output=True, timeout=120)
    assert p.returncode == 255 and out.read_bytes() == b"precious\n"
    p = subprocess.run([binary, "-i", str(fin), "-o", str(out), "-x", "ont", "-q", "10"], capture_output=True, timeout=120)
    assert p.returncode == 0 and out.read_bytes().startswith(b"@")


def test_cli_gz_output_and_fasta(binary, golden_dir, tmp_path):
    """-o *.fq.gz (per-record gzip members) inflates to the reference's output; -o *.fa keeps the bases."""
    import gzip
    import json
    name = "ont_zoo"
    cmd = json.load(open(os.path.join(golden_dir, name + ".cmd.json")))
    ref_out = gzip.open(os.path.join(golden_dir, name + ".out.fq.gz"), "rb").read()
    fin = tmp_path / "in.fq"
    fin.write_bytes(gzip.open(os.path.join(golden_dir, name + ".in.fq.gz"), "rb").read())
    fa = tmp_path / "ad.fa"
    fa.write_text("".join(">a%d\n%s\n" % (i, a) for i, a in enumerate(cmd["adapters"])))
    base = [binary, "-i", str(fin), "-a", str(fa)] + cmd["flags"].split()
    subprocess.run(base + ["-o", str(tmp_path / "o.fq.gz")], check=True, capture_output=True)
    assert gzip.open(tmp_path / "o.fq.gz", "rb").read() == ref_out
    subprocess.run(base + ["-o", str(tmp_path / "o.fa")], check=True, capture_output=True)
    ref_lines = ref_out.split(b"\n")
    exp = b"".join(b">" + ref_lines[i][1:] + b"\n" + ref_lines[i + 1] + b"\n" for i in range(0, len(ref_lines) - 3, 4))
    assert (tmp_path / "o.fa").read_bytes() == exp
    # gz input
    gz_in = tmp_path / "in2.fq.gz"
    gz_in.write_bytes(open(os.path.join(golden_dir, name + ".in.fq.gz"), "rb").read())
    subprocess.run([binary, "-i", str(gz_in), "-a", str(fa)] + cmd["flags"].split() + ["-o", str(tmp_path / "o2.fq")],
                   check=True, capture_output=True)
    assert (tmp_path / "o2.fq").read_bytes() == ref_out


@pytest.mark.parametrize("name", ["ont_zoo", "hifi_auto", "down_r"])
def test_cli_several_devices(binary, golden_dir, name):
    """--devices a,b,c: one context and feeder thread per entry (here three on the same device), batches
    re-sequenced by the writer, tallies merged on the host: everything equals the single-context run."""
    cli_check.run_case(binary, golden_dir, name, extra_args=["--devices", "0,0,0"])


@pytest.mark.parametrize("name,block", [("ont_zoo", "1000"), ("hifi_auto", "4096"), ("ont_fasta", "777"), ("down_r", "100000")])
def test_cli_parallel_line_scan(binary, golden_dir, name, block, monkeypatch):
    """Newlines located ahead of the parser by several threads, block by block (tiny blocks here: lines span
    many of them): same records, same everything."""
    monkeypatch.setenv("TGSF_SCAN_THREADS", "3")
    monkeypatch.setenv("TGSF_SCAN_BLOCK", block)
    cli_check.run_case(binary, golden_dir, name)


@pytest.mark.parametrize("name", ["ont_zoo", "hifi_zoo", "hifi_bam", "ont_fasta"])
@pytest.mark.parametrize("writer", ["auto", "writev"])
def test_cli_threaded_pipeline(binary, golden_dir, name, writer, monkeypatch):
    """-t 8 with tiny batches: three contexts / feeders on the device, batches re-sequenced by the planner, the
    records of every batch copied into the mapped output file by several fill threads (or gathered by the
    single-stream writer): byte-equal output, same stderr, same report."""
    monkeypatch.setenv("TGSF_BATCH_BYTES", "30000")
    monkeypatch.setenv("TGSF_CTX_PER_DEVICE", "3")
    monkeypatch.setenv("TGSF_FILL_MIN_BYTES", "1")
    monkeypatch.setenv("TGSF_SCAN_BLOCK", "5000")
    monkeypatch.setenv("TGSF_WRITER", writer)
    monkeypatch.setenv("TGSF_STRIDE_BYTES", "70000")              # the output file grows stride by stride
    cli_check.run_case(binary, golden_dir, name, extra_args=["-t", "8"])


@pytest.mark.parametrize("name,compress", [("ont_zoo", "gzip"), ("ont_zoo", "bgzf"), ("hifi_auto", "gzip"), ("ont_fasta", "bgzf"),
                                           ("hifi_fasta_auto", "gzip"), ("hifi_bam", None), ("hifi_bam_auto", None), ("ont_sam", None),
                                           ("ont_repeat", "bgzf"), ("huge_adapter", "gzip")])
@pytest.mark.parametrize("chunk", ["20000", "3000000"])
def test_cli_streamed_input(binary, golden_dir, name, compress, chunk, monkeypatch):
    """Compressed / BAM / SAM input decoded piece by piece in bounded memory (textsource.h), forced here on small
    files with tiny chunks (records larger than a chunk, lines cut by chunk and gzip-member ends): same output, same
    stderr, same report as the whole-file way and as the reference."""
    monkeypatch.setenv("TGSF_STREAM_MIN_BYTES", "1")
    monkeypatch.setenv("TGSF_CHUNK_BYTES", chunk)
    monkeypatch.setenv("TGSF_FILL_MIN_BYTES", "1")
    cli_check.run_case(binary, golden_dir, name, extra_args=["-t", "4"], compress=compress)


@pytest.mark.parametrize("cap,zlib_only", [("3000", None), ("200000", None), (None, "1")])
def test_cli_gzip_member_paths(binary, golden_dir, cap, zlib_only, monkeypatch):
    """gzip members are decoded whole by libdeflate while they fit its buffer limit, by zlib's streaming inflater
    otherwise (forced here: every member too large / only the first one / libdeflate not used at all) -- same text."""
    monkeypatch.setenv("TGSF_STREAM_MIN_BYTES", "1")
    monkeypatch.setenv("TGSF_CHUNK_BYTES", "50000")
    if cap:
        monkeypatch.setenv("TGSF_GZ_MEMBER_CAP", cap)
    if zlib_only:
        monkeypatch.setenv("TGSF_ZLIB_INPUT", zlib_only)
    cli_check.run_case(binary, golden_dir, "ont_zoo", extra_args=["-t", "4"], compress="gzip")


@pytest.mark.parametrize("stream_min", ["1", "999999999"])
def test_cli_corrupt_gzip(binary, golden_dir, tmp_path, stream_min, monkeypatch):
    """A damaged .gz ends the run with the reference's message (:636) and exit status, streamed or decoded whole."""
    import gzip
    raw = gzip.open(os.path.join(golden_dir, "ont_zoo.in.fq.gz"), "rb").read()
    blob = bytearray(gzip.compress(raw, 4))
    for k in range(len(blob) // 2, len(blob) // 2 + 64):
        blob[k] ^= 0x5A
    f = tmp_path / "bad.fq.gz"
    f.write_bytes(bytes(blob))
    monkeypatch.setenv("TGSF_STREAM_MIN_BYTES", stream_min)
    p = subprocess.run([binary, "-i", str(f), "-o", str(tmp_path / "o.fq"), "-x", "ont", "-t", "2"], capture_output=True)
    assert p.returncode == 255 and b"Error encountered while decompressing file" in p.stderr


@pytest.mark.parametrize("name", ["ont_zoo", "hifi_bam", "down_r"])
def test_cli_detached_exit(binary, golden_dir, name, monkeypatch):
    """TGSF_DETACH=1: the work happens in a child process whose address space is taken down in the background after the
    status is out -- same results as the default (one process, mappings dropped piece by piece during the run,
    everything on the caller's clock)."""
    monkeypatch.setenv("TGSF_DETACH", "1")
    monkeypatch.setenv("TGSF_BATCH_BYTES", "30000")
    monkeypatch.setenv("TGSF_FILL_MIN_BYTES", "1")
    cli_check.run_case(binary, golden_dir, name, extra_args=["-t", "4"])


def test_cli_exit_status_comes_through_the_parent(binary, tmp_path, monkeypatch):
    """(TGSF_DETACH=1) Fatal paths of the working child (here: an output file that cannot be opened, after threads exist) reach the caller
    as the reference's exit status, and the caller's pipes close when the work is done."""
    monkeypatch.setenv("TGSF_DETACH", "1")
    f = tmp_path / "a.fq"
    f.write_bytes(b"@r\n" + b"ACGT" * 400 + b"\n+\n" + b"I" * 1600 + b"\n")
    p = subprocess.run([binary, "-i", str(f), "-x", "ont", "-o", str(tmp_path / "no_such_dir" / "o.fq")], capture_output=True, timeout=120)
    assert p.returncode == 1 and b"Failed to open file" in p.stderr
    p = subprocess.run([binary, "-i", str(f), "-x", "ont", "-o", str(tmp_path / "o.fq")], capture_output=True, timeout=120)
    assert p.returncode == 0 and (tmp_path / "o.fq").read_bytes().startswith(b"@r\n")


def test_cli_usage_and_errors(binary, tmp_path):
    p = subprocess.run([binary], capture_output=True)
    assert p.returncode == 1 and p.stdout.startswith(b"Usage: tgsfilter -i TGS.raw.fq.gz -x ont -o TGS.clean.fq.gz")
    p = subprocess.run([binary, "-i", "/nonexistent.fq", "-x", "ont"], capture_output=True)
    assert p.returncode != 0 and b"Error: Can't find this file for -i /nonexistent.fq" in p.stderr
    f = tmp_path / "a.fq"
    f.write_bytes(b"@r\nACGT\n+\nIIII\n")
    p = subprocess.run([binary, "-i", str(f), "-z", "1"], capture_output=True)
    assert p.returncode == 1 and b"Error: UnKnow argument -z" in p.stderr
    p = subprocess.run([binary, "-i", str(f), "-l"], capture_output=True)
    assert p.returncode == 1 and b"Error: Lack Argument for [ -l ]" in p.stderr
    p = subprocess.run([binary, "-i", str(f), "-o", "x.fq"], capture_output=True)
    assert b"Error: lack argument for the must: -x" in p.stderr
