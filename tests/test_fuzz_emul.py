"""Seeded parameter fuzzing of the kernel logic through the serial emulation (CPU)."""
import pytest

from tests import fuzz
from tests.test_emul_parity import emul  # noqa: F401  (fixture)


@pytest.mark.parametrize("seed", range(100, 160))
def test_fuzz_emul(emul, seed):  # noqa: F811
    fuzz.run_case(emul, seed, 24)
