"""-m gpu: seeded parameter fuzzing of the HIP path against the oracle."""
import pytest

from tests import fuzz

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(100, 160))
def test_fuzz_gpu(seed):
    fuzz.run_case(None, seed, 120)
