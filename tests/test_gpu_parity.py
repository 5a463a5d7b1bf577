"""-m gpu: the HIP path (tgsfilter_amd/libtgsf.so through the C ABI) against the oracle and the goldens."""
import numpy as np
import pytest

from tests import hostmodel, parity
from tgsfilter_amd import abi, capi, synth

pytestmark = pytest.mark.gpu


def test_gpu_edlib_vectors(golden_dir):
    parity.edlib_vectors(None, golden_dir)


@pytest.mark.parametrize("name", hostmodel.GOLDEN_CASES)
def test_gpu_golden(golden_dir, name):
    parity.golden_case(None, golden_dir, name)


def test_gpu_unaligned_offsets():
    reads = synth.make_reads(5, 200, "ont", mean_len=2500, zoo=True, pmid=0.1)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0,
                                     head_trim=7, tail_trim=3), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads, align=1, explicit_lengths=False)
    ctx.close()


@pytest.mark.parametrize("kind,n,mean_len,ads", [
    ("ont", 300, 9000, [synth.ONT_RAPID, synth.ONT_RAPID_RC]),
    ("hifi", 300, 6000, [synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC]),
    ("ont", 60, 30000, [synth.ONT_RAPID, synth.ONT_RAPID_RC, synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC]),
])
def test_gpu_random_batches(kind, n, mean_len, ads):
    reads = synth.make_reads(21, n, kind, mean_len=mean_len, zoo=True, pmid=0.05)
    p = parity.sized(abi.make_params(kind, adapters=ads, min_q=10.0 if kind == "ont" else 20.0), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


def test_gpu_counters_accumulate_over_batches():
    """Two batches on one context == the oracle run over both (tallies are additive)."""
    from oracle import orc
    r1 = synth.make_reads(31, 80, "ont", mean_len=5000, zoo=True)
    r2 = synth.make_reads(32, 80, "ont", mean_len=7000, zoo=True)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC]), r1 + r2)
    ctx = capi.Context(p, 0)
    exp = np.zeros(ctx.ctr_words, dtype=np.uint64)
    for rr in (r1, r2):
        seq, qual, off, ln = synth.pack(rr)
        ctx.submit(seq, qual, off[:-1].copy(), ln)
        orc.filter_batch(p, seq, qual, off, ln, n_bins=ctx.n_bins, ctr=exp)
    assert np.array_equal(ctx.counters(), exp)
    ctx.reset_counters()
    assert not ctx.counters().any()
    ctx.close()
