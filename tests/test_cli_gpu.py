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
