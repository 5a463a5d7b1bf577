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
