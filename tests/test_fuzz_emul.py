"""Seeded parameter fuzzing of the kernel logic through the serial emulation (CPU)."""
import pytest

from tests import fuzz
from tests.test_emul_parity import emul  # noqa: F401  (fixture)


@pytest.mark.parametrize("seed", range(100, 160))
def test_fuzz_emul(emul, seed):  # noqa: F811
    fuzz.run_case(emul, seed, 24)


@pytest.mark.parametrize("seed", [20081, 20104])
def test_fuzz_emul_loose_thresholds_long_adapters(emul, seed, monkeypatch):  # noqa: F811
    """Cases a wider campaign found: -M 20..25 with 150- and 241-bp adapters makes every lane list candidates (the first lane
    of a read one per new low while the score comes down from Q); the candidate pool has room for that."""
    monkeypatch.setenv("TGSF_FUZZ_GATE_P", "0.6")
    fuzz.run_case(emul, seed, 150)


@pytest.mark.parametrize("seed", [30187, 30322, 30070])
def test_fuzz_emul_loose_thresholds_long_reads(emul, seed, monkeypatch):  # noqa: F811
    """-M 1 / -M 20 on long reads: every lane of the middle scan has a best value at or below k; lanes whose best is worse than
    what the read has handed over so far drop their columns (the GPU suite runs the 70-kb cases the campaign found)."""
    monkeypatch.setenv("TGSF_FUZZ_GATE_P", "0.8")
    monkeypatch.setenv("TGSF_FUZZ_MEAN_LEN", "20000")
    fuzz.run_case(emul, seed, 8)
