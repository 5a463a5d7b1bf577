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


@pytest.mark.parametrize("name", hostmodel.GOLDEN_CASES)
def test_cli_golden(binary, golden_dir, name):
    cli_check.run_case(binary, golden_dir, name)
