"""-m gpu: seeded parameter fuzzing of the HIP path against the oracle."""
import pytest

from tests import fuzz

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(100, 160))
def test_fuzz_gpu(seed):
    fuzz.run_case(None, seed, 120)


@pytest.mark.parametrize("seed", [20081, 20104])
def test_fuzz_gpu_loose_thresholds_long_adapters(seed, monkeypatch):
    """Cases a wider campaign found: -M 20..25 with 150- and 241-bp adapters makes every lane list candidates (the first lane
    of a read one per new low while the score comes down from Q); the candidate pool has room for that."""
    monkeypatch.setenv("TGSF_FUZZ_GATE_P", "0.6")
    fuzz.run_case(None, seed, 150)


@pytest.mark.parametrize("seed", [30187, 30322])
def test_fuzz_gpu_loose_thresholds_long_reads(seed, monkeypatch):
    """Found by the long-read campaign: -M 1 (k = Q-1) on 70-kb reads -- every lane of the middle scan has a best value at or
    below k.  Lanes whose best is worse than what the read has handed over so far drop their columns (mid_best)."""
    monkeypatch.setenv("TGSF_FUZZ_GATE_P", "0.8")
    monkeypatch.setenv("TGSF_FUZZ_MEAN_LEN", "70000")
    fuzz.run_case(None, seed, 30)
