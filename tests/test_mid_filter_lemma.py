"""The claim behind the filtering middle scan (k_mid_flat<AT, Hot32, FS>), checked on its own, away from the kernels:
for every text column j, the bottom-row value of the adapter's LAST 32 rows (infix search, free start) is never above the whole
adapter's -- so every column the whole adapter reaches within k is a column its last 32 rows reach within k -- and neighbouring
bottom-row values differ by at most 1 (why looking at every 2nd column against k + 1 loses nothing).  Plain dynamic programming
here, not the bit-parallel column, on random and on low-complexity texts with planted, mutated adapters."""
import numpy as np
import pytest

from tgsfilter_amd import synth

ADAPTERS = [synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC, synth.ONT_RAPID, b"AAAAAAAAAAAAAAAAAATTAACGGAGGAGGAGGA",
            b"AGCAATACGTAACTGAACGAAGTACAGGAAAAAAAA", b"GGAACCTCTCTGACTTGGAACCTCTCTGACAAAAAGGTTAAACACCCAAGCAGACGCCAGCAAT"]


def bottom_row(q: bytes, t: bytes) -> np.ndarray:
    """D[len(q)][j] for every column j of t: edit distance of q to the best substring of t ENDING at j (edlib's HW mode)."""
    qa = np.frombuffer(q, dtype=np.uint8)
    col = np.arange(len(q) + 1)
    out = np.empty(len(t), dtype=np.int64)
    for j, ch in enumerate(t):
        new = np.empty_like(col)
        new[0] = 0
        diag = col[:-1] + (qa != ch)
        up = col[1:] + 1
        best = np.minimum(diag, up)
        # new[i] = min(best[i-1], new[i-1] + 1): a running minimum of best[m] + (i - 1 - m)
        idx = np.arange(len(q))
        new[1:] = np.minimum.accumulate(np.concatenate(([0], best)) - np.arange(len(q) + 1))[1:] + idx + 1
        new[1:] = np.minimum(new[1:], best)
        col = new
        out[j] = col[-1]
    return out


def texts(rng, adapter):
    t = bytearray(synth._ACGT[rng.integers(0, 4, 3000)].tobytes())
    for at, rate in ((200, 0.0), (700, 0.1), (1300, 0.2), (2000, 0.3)):
        a = synth.mutate(rng, adapter, rate)
        t[at:at + len(a)] = a
    yield bytes(t)
    yield (b"T" * 400 + b"A" * 400 + b"TC" * 300 + b"GGA" * 200 + b"AAC" * 200)


@pytest.mark.parametrize("k_ad", range(len(ADAPTERS)))
def test_last_32_rows_never_above_the_whole_adapter(k_ad):
    rng = np.random.default_rng(500 + k_ad)
    ad = ADAPTERS[k_ad]
    assert len(ad) > 32
    for t in texts(rng, ad):
        whole, tail = bottom_row(ad, t), bottom_row(ad[-32:], t)
        assert (tail <= whole).all()
        assert (np.abs(np.diff(tail)) <= 1).all() and (np.abs(np.diff(whole)) <= 1).all()
        for k in (1, 2, 6, 11, 12):
            hit = np.flatnonzero(whole <= k)
            # stride 1: the column itself; stride 2: an odd column of the same 16-column chunk within k + 1
            assert (tail[hit] <= k).all()
            for j in hit:
                odd = j | 1
                assert odd < len(t) and tail[odd] <= k + 1 or odd >= len(t)
