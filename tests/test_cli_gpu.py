"""-m gpu: the real command line (tgsfilter_amd/bin/tgsfilter, linked against the HIP library) end to end
against everything the reference produced for the golden cases."""
import os

import pytest

from tests import cli_check, hostmodel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
