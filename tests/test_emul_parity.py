"""Kernel-logic parity on a GPU-less box: tests/emul/libtgsf_emul.so is the SAME kernel source
(tgsfilter_amd/csrc) compiled with -DTGSF_EMUL and executed lane by lane on the CPU.  It checks
the kernels' decomposition (tiles, segments with warm-up, candidate lists, region merging) against
the oracle before GPU time is spent; the -m gpu tests repeat these checks on the real HIP build."""
import os
import subprocess

import numpy as np
import pytest

from tests import hostmodel, parity
from tgsfilter_amd import abi, capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL = os.environ.get("TGSF_EMUL_LIB") or os.path.join(EMUL_DIR, "libtgsf_emul.so")     # (tests/manual/sanitize_emul.py: the sanitizer build)


@pytest.fixture(scope="module")
def emul():
    subprocess.run(["make", "-s", "-C", EMUL_DIR], check=True)
    return EMUL


def test_emul_edlib_vectors(emul, golden_dir):
    parity.edlib_vectors(emul, golden_dir)


@pytest.mark.parametrize("name", ["ont_zoo", "ont_trim", "ont_discard", "hifi_zoo", "long_adapter", "qc_only", "ont_m1", "huge_adapter", "ont_phred64", "ont_e1300", "wide_adapter"])
def test_emul_golden(emul, golden_dir, name):
    parity.golden_case(emul, golden_dir, name)


def test_emul_unaligned_offsets(emul):
    """Tightly packed CSR (no padding, implicit lengths): every misalignment of tile starts."""
    reads = synth.make_reads(5, 40, "ont", mean_len=2500, zoo=True, pmid=0.1)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0,
                                     head_trim=7, tail_trim=3), reads)
    ctx = capi.Context(p, 0, emul)
    parity.compare_batch(ctx, p, reads, align=1, explicit_lengths=False)
    ctx.close()


def test_emul_long_reads_and_four_adapters(emul):
    """Reads spanning several stats tiles and several middle segments; 4 adapters in one pass."""
    reads = synth.make_reads(6, 12, "ont", mean_len=15000, zoo=True, pmid=0.3)
    ads = [synth.ONT_RAPID, synth.ONT_RAPID_RC, synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC]
    p = parity.sized(abi.make_params("ont", adapters=ads, min_q=8.0), reads)
    ctx = capi.Context(p, 0, emul)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


def test_emul_edge_lengths(emul):
    from tests.test_gpu_parity import _edge_reads
    reads = _edge_reads()
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=7.0,
                                     min_len=100, head_trim=3, tail_trim=2), reads)
    ctx = capi.Context(p, 0, emul)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


@pytest.mark.parametrize("kind,align", [("ont", 16), ("ont", 1), ("hifi", 16)])
def test_emul_quality_bytes_of_128_and_above(emul, kind, align):
    """Such a byte stands for its value - 256, as in the reference's arithmetic on a signed char (src/TGSFilter.cpp:1455-1457)."""
    reads = parity.high_quality_byte_reads(kind=kind)
    p = parity.sized(abi.make_params(kind, adapters=[synth.ONT_RAPID if kind == "ont" else synth.PACBIO_BLUNT], min_q=7.0, head_trim=13, tail_trim=4), reads)
    ctx = capi.Context(p, 0, emul)
    parity.compare_batch(ctx, p, reads, align=align)
    ctx.close()
    p.max_batch_bases = 2 * sum(len(r[1]) for r in reads) + 64 * len(reads) + 4096      # the FASTQ text itself as the batch
    ctx = capi.Context(p, 0, emul)
    parity.compare_batch_in_place(ctx, p, reads)
    ctx.close()


def test_emul_loose_thresholds(emul):
    """-M far below the default: most columns are within k, every lane records ties all the time."""
    reads = synth.make_reads(8, 16, "ont", mean_len=5000, zoo=True, pmid=0.3)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=8.0,
                                     mid_match_len=18, end_match_len=8, mid_sim=0.8, end_sim=0.7), reads)
    ctx = capi.Context(p, 0, emul)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


def test_emul_in_place_fastq_text(emul):
    """Both streams read in place from the raw FASTQ text (separate seq / qual offsets, any alignment)."""
    reads = synth.make_reads(12, 40, "ont", mean_len=3000, zoo=True, pmid=0.1)
    p = abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0, head_trim=4)
    p.max_batch_reads = len(reads)
    p.max_batch_bases = 2 * sum(len(r[1]) for r in reads) + 64 * len(reads) + 4096
    p.max_read_len = max(len(r[1]) for r in reads)
    ctx = capi.Context(p, 0, emul)
    parity.compare_batch_in_place(ctx, p, reads)
    ctx.close()


@pytest.mark.parametrize("pval,k", [(300, 11), (40, 9), (2000, 12), (40, 15), (30, 21), (4000, 32)])
def test_emul_repeat_gate(emul, pval, k):
    reads = parity.repeat_reads()
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=10.0,
                                     min_repeat=pval, kmer=k), reads)
    ctx = capi.Context(p, 0, emul)
    res, frags, ctr = parity.compare_batch(ctx, p, reads)
    assert (frags["flags"] & abi.FF_REPEAT).any() and (frags["flags"] & abi.FF_PASS).any()
    ctx.close()


@pytest.mark.parametrize("k,lens", [(11, "TINY"), (11, "SHORT"), (11, "LONG"), (9, "LONG"), (10, "SHORT"), (12, "SHORT"), (13, "TINY"),
                                    (13, "LONG"), (5, "SHORT"), (3, "TINY"), (1, "SHORT")])
def test_emul_repeat_gate_exact_counts(emul, k, lens):
    """Reads built to have repeat == T and == T-1 exactly, on the chunk and window seams of k_repeat."""
    parity.repeat_threshold_case(emul, k, getattr(parity, "REPEAT_" + lens), max_runs=16)


@pytest.mark.parametrize("k,lens", [(14, "SHORT"), (16, "LONG"), (16, "TINY"), (17, "SHORT"), (20, "LONG"), (31, "SHORT"), (31, "TINY")])
def test_emul_repeat_gate_exact_counts_keys(emul, k, lens):
    """The same through k_repeat_keys (32-bit keys to k = 16, 64-bit above)."""
    parity.repeat_threshold_case(emul, k, getattr(parity, "REPEAT_" + lens), max_runs=10)


@pytest.mark.parametrize("k", [14, 20])
def test_emul_repeat_gate_skewed_composition(emul, k):
    """Long reads over two letters: a pass's table fills up and the fragment starts over with more passes."""
    parity.repeat_threshold_case(emul, k, parity.REPEAT_SKEW, max_runs=4, alphabet=b"AC")


@pytest.mark.parametrize("k,plant", [(13, 0), (16, 0), (16, 95), (22, 140)])
def test_emul_repeat_gate_pass_seams(emul, k, plant):
    """k_repeat_keys around its pass sizes (131 072 k-mers a pass: one pass, two, four), on random reads whose first scan
    flags tens to thousands of first occurrences: -p below that number needs the exact count of the second scan, -p above
    it drops the fragment without one; with k-mers copied within the read the count sits on either side of -p."""
    parity.repeat_threshold_case(emul, k, parity.REPEAT_SHARE, max_runs=6, share=True, plant=plant, extra_thresholds=(40, 100, 2500))


# The first middle scan of a batch: k_mid_scan1 (TGSF_MID_FLAT=0) and k_mid_flat under stretch schedules from one chunk
# per stretch (every read cut into hundreds of stretches, each with its warm-up) to the default
MID_SCAN_ENVS = [{"TGSF_MID_FLAT": "0"},
                 {"TGSF_FLAT_PMIN": "1", "TGSF_FLAT_PMAX": "1", "TGSF_FLAT_F0": "128"},
                 {"TGSF_FLAT_PMIN": "1", "TGSF_FLAT_PMAX": "8", "TGSF_FLAT_F0": "100"},
                 {"TGSF_FLAT_PMIN": "4", "TGSF_FLAT_PMAX": "64", "TGSF_FLAT_F0": "250"},
                 {"TGSF_FLAT_PMIN": "16", "TGSF_FLAT_PMAX": "256", "TGSF_FLAT_F0": "224"}]


@pytest.mark.parametrize("env", MID_SCAN_ENVS, ids=lambda e: "-".join(f"{k[5:].lower()}{v}" for k, v in e.items()))
def test_emul_mid_scan_variants(emul, golden_dir, env):
    parity.mid_scan_variants(emul, golden_dir, env)


MID_FILTER_ENVS = [{"TGSF_MID_FILTER": "0"}, {"TGSF_MID_FILTER": "1"}, {"TGSF_MID_FILTER": "2"},
                   {"TGSF_MID_FILTER": "2", "TGSF_FLAT_PMIN": "1", "TGSF_FLAT_PMAX": "4", "TGSF_FLAT_F0": "128"},
                   {"TGSF_MID_FILTER": "1", "TGSF_RECHECK_CAP": "5"}]        # (a list of five marks: the rest rechecked where found)


@pytest.mark.parametrize("env", MID_FILTER_ENVS, ids=lambda e: "-".join(f"{k[5:].lower()}{v}" for k, v in e.items()))
def test_emul_mid_filter(emul, env):
    parity.mid_filter_cases(emul, env)


@pytest.mark.parametrize("mode", ["direct", "difference"])
def test_emul_clean_table_strategy(emul, golden_dir, mode):
    parity.clean_table_strategy(emul, mode, golden_dir)


@pytest.mark.parametrize("mode", [None, "direct", "difference"])
def test_emul_no_qual(emul, mode):
    parity.no_qual_batch(emul, mode)


def test_emul_submit_async(emul):
    parity.async_two_contexts(emul)


def test_emul_align_windows_random(emul):
    parity.align_windows_random(emul, 1500)


def test_emul_more_than_64_drop_regions(emul):
    parity.many_regions(emul)


def _tie_reads():
    """Reads whose middle-scan minimum is tied column after column: homopolymers against homopolymer
    adapters, a (CT)n read against the PacBio blunt adapter, beside ordinary reads."""
    rng = np.random.default_rng(77)
    q = lambda n: bytes((rng.integers(15, 35, n) + 33).astype(np.uint8))
    reads = synth.make_reads(9, 6, "ont", mean_len=3000, zoo=True, pmid=0.5)
    reads.append((b"polyA", b"A" * 100000, q(100000)))      # 100 000 tied columns: beyond the 65 536 + ... slots of a small context
    reads.append((b"polyT_ends", b"ACGT" * 100 + b"T" * 9000 + b"GATTACA" * 60, q(400 + 9000 + 420)))
    reads.append((b"ct", b"CT" * 6000, q(12000)))
    reads.append((b"polyA_short", b"A" * 700, q(700)))
    return reads


@pytest.mark.parametrize("order", ["forward", "reverse"])
@pytest.mark.parametrize("adapters,m_mid", [([b"A" * 50, b"T" * 50], 35), ([synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC], 1),
                                            ([b"A" * 50, b"T" * 50, b"AC" * 45 + b"G"], 35)])
def test_emul_candidate_pool_overflow_is_handled(emul, adapters, m_mid, order, monkeypatch, capfd):
    """With the sizing hints a minimal caller passes, reads whose minimum is tied everywhere used to end in TGSF_E_CAPACITY
    ("candidate pool overflow"); the reference completes them (include/edlib.cpp:660-672: every column at the global
    minimum is a location).  The library now re-runs the scan with a pool that fits: same records, same tallies as the oracle."""
    monkeypatch.setenv("TGSF_TRACE_POOL", "1")
    monkeypatch.setenv("TGSF_EMUL_ORDER", order)          # reverse: the emulated lanes run last to first (a GPU's order is its own)
    reads = _tie_reads()
    p = parity.sized(abi.make_params("ont", adapters=adapters, min_q=7.0, mid_match_len=m_mid, end_match_len=4), reads)
    ctx = capi.Context(p, 0, emul)
    parity.compare_batch(ctx, p, reads)
    err = capfd.readouterr().err
    assert "candidate pool overflow" in err and "(grown)" in err      # the fallback DID run, and had to grow the pool
    ctx.close()


@pytest.mark.parametrize("order", ["forward", "reverse"])
@pytest.mark.parametrize("name", ["ont_zoo", "hifi_zoo", "ont_m1", "huge_adapter"])
def test_emul_golden_through_the_overflow_path(emul, golden_dir, name, order, monkeypatch, capfd):
    """A pool of 2 slots: every batch with more than two candidates takes the count-and-rescan path; the goldens must not notice."""
    monkeypatch.setenv("TGSF_POOL_CAP", "2")
    monkeypatch.setenv("TGSF_TRACE_POOL", "1")
    monkeypatch.setenv("TGSF_EMUL_ORDER", order)
    parity.golden_case(emul, golden_dir, name)
    assert "candidate pool overflow" in capfd.readouterr().err


def test_emul_pool_overflow_over_several_batches(emul, monkeypatch):
    """Tallies accumulate correctly when some batches of a run take the overflow path and others do not; a batch submitted
    asynchronously is completed by tgsf_wait."""
    monkeypatch.setenv("TGSF_POOL_CAP", "3")
    plain = synth.make_reads(21, 10, "ont", mean_len=2500, zoo=False, pmid=0.0, p5=0.0)
    busy = synth.make_reads(22, 10, "ont", mean_len=2500, zoo=True, pmid=1.0)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=7.0), plain + busy)
    ctx = capi.Context(p, 0, emul)
    exp = np.zeros(ctx.ctr_words, dtype=np.uint64)
    from oracle import orc
    for reads in (plain, busy, plain, busy):
        seq, qual, offsets, lengths = synth.pack(reads)
        got_r, got_f = ctx.submit(seq, qual, offsets[:-1].copy(), lengths)
        exp_r, exp_f, exp = orc.filter_batch(p, seq, qual, offsets, lengths, n_bins=ctx.n_bins, ctr=exp)
        assert np.array_equal(got_r, exp_r) and np.array_equal(got_f, exp_f)
    assert np.array_equal(ctx.counters(), exp)
    ctx.close()


def test_emul_pool_overflow_in_batches_enqueued_together(emul, monkeypatch):
    """tgsf_submit_device, several batches enqueued before one tgsf_wait (in the emulation device memory is host memory):
    the ones whose candidate pool overflows are left alone by their first run -- the fragment count says so -- and run
    again by tgsf_wait from their inputs, the others are untouched; records and tallies equal the oracle's (what the first
    runs added to the raw tables is not added twice)."""
    monkeypatch.setenv("TGSF_POOL_CAP", "3")
    from oracle import orc
    plain = synth.make_reads(21, 10, "ont", mean_len=2500, zoo=False, pmid=0.0, p5=0.0)
    busy = synth.make_reads(22, 10, "ont", mean_len=2500, zoo=True, pmid=1.0)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=7.0), plain + busy)
    ctx = capi.Context(p, 0, emul)
    exp = np.zeros(ctx.ctr_words, dtype=np.uint64)
    keep, want = [], []
    for reads in (busy, plain, busy, busy, plain):
        seq, qual, offsets, lengths = synth.pack(reads)
        off = offsets[:-1].astype(np.uint64).copy()
        exp_r, exp_f, exp = orc.filter_batch(p, seq, qual, offsets, lengths, n_bins=ctx.n_bins, ctr=exp)
        o_r = np.zeros(len(reads), dtype=abi.READ_RESULT_DTYPE)
        o_f = np.zeros(len(exp_f) + 16, dtype=abi.FRAGMENT_DTYPE)
        o_n = np.zeros(1, dtype=np.uint32)
        ln = lengths.astype(np.uint32).copy()
        keep.append((seq, qual, off, ln, o_r, o_f, o_n))
        want.append((exp_r, exp_f))
        ctx.submit_device(seq.ctypes.data, qual.ctypes.data, off.ctypes.data, ln.ctypes.data, len(reads), seq.size,
                          o_r.ctypes.data, o_f.ctypes.data, len(o_f), o_n.ctypes.data, None)
    flagged = [int(k[6][0]) == abi.NFRAGS_NOT_FINAL for k in keep]
    assert flagged == [True, False, True, True, False]
    ctx.wait()
    for (seq, qual, off, ln, o_r, o_f, o_n), (exp_r, exp_f) in zip(keep, want):
        assert np.array_equal(o_r, exp_r) and np.array_equal(o_f[:int(o_n[0])], exp_f)
    assert np.array_equal(ctx.counters(), exp)
    ctx.close()


@pytest.mark.parametrize("name", ["ont_zoo", "hifi_zoo", "ont_discard", "long_adapter"])
def test_emul_golden_with_lanes_in_reverse_order(emul, golden_dir, name, monkeypatch):
    """The candidate lists of the one-pass scan are in whatever order the lanes append: results must not depend on it."""
    monkeypatch.setenv("TGSF_EMUL_ORDER", "reverse")
    parity.golden_case(emul, golden_dir, name)


ONT_LIGATION_28 = b"AATGTACTTCGTTCAGTTACGTATTGCT"      # src/TGSFilter.cpp:2974-2977 (library entries 4..7)
ONT_LIGATION_28_RC = b"AGCAATACGTAACTGAACGAAGTACATT"
ONT_LIGATION_22 = b"GCAATACGTAACTGAACGAAGT"
ONT_LIGATION_22_RC = b"ACTTCGTTCAGTTACGTATTGC"


@pytest.mark.parametrize("ads", [[ONT_LIGATION_28, ONT_LIGATION_28_RC], [ONT_LIGATION_22, ONT_LIGATION_22_RC, ONT_LIGATION_28, ONT_LIGATION_28_RC],
                                 [ONT_LIGATION_22, ONT_LIGATION_22_RC, synth.ONT_RAPID, synth.ONT_RAPID_RC], [b"ACGTTGCA" * 4, ONT_LIGATION_22],
                                 [ONT_LIGATION_28, ONT_LIGATION_22_RC, b"ACGTTGCA" * 4]])      # three: passes of two and one
@pytest.mark.parametrize("no32", ["0", "1"])
def test_emul_short_adapters_dword_column(emul, ads, no32, monkeypatch):
    """Adapters of at most 32 bp run the middle scan with the one-dword column (Hot32): same locations as the 64-bit column
    (TGSF_NO_HOT32=1) and as the oracle; mixed sets take one pass per word class."""
    monkeypatch.setenv("TGSF_NO_HOT32", no32)
    reads = synth.make_reads(31, 30, "ont", mean_len=5000, zoo=True, pmid=0.8, adapter=ads[0], err=0.06)
    p = parity.sized(abi.make_params("ont", adapters=ads, min_q=7.0, mid_match_len=14, end_match_len=4), reads)
    ctx = capi.Context(p, 0, emul)
    r, f, _ = parity.compare_batch(ctx, p, reads)
    assert (r["flags"] & abi.RF_ADMID).any()              # the planted copies were found in the middle
    ctx.close()


def _shared_prefix_read(k=31, units=9000, seed=5):
    """Thousands of distinct duplicated k-mers that share their first 16 bases (no number of leading bases
    separates them into passes whose table holds them -- the passes of k_repeat_keys own the keys by a hash of the whole key)."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    pre = acgt[rng.integers(0, 4, 16)].tobytes()
    block = b"".join(pre + acgt[rng.integers(0, 4, k - 16)].tobytes() for _ in range(units))
    s = block + block                                   # every k-mer of the block occurs twice
    q = bytes((rng.integers(15, 35, len(s)) + 33).astype(np.uint8))
    return (b"shared_prefix", s, q)


def test_emul_repeat_gate_shared_prefix_fragment(emul):
    """k = 31, 279 000 distinct duplicated k-mers in one fragment: the keys kernel starts over with as many passes as its table
    needs (8 -> 128); the count (checked with numpy on both sides of the gate) stays exact."""
    read = _shared_prefix_read()
    c = parity._kmer_repeat_np(read[1], 31)
    assert c > 250_000
    for pval, kept in ((c, True), (c + 1, False)):
        p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID], min_q=7.0, min_repeat=pval, kmer=31), [read])
        ctx = capi.Context(p, 0, emul)
        r, f = ctx.submit(*[x if i != 2 else x[:-1].copy() for i, x in enumerate(synth.pack([read]))])
        assert len(f) == 1 and bool(f["flags"][0] & abi.FF_PASS) == kept and bool(f["flags"][0] & abi.FF_REPEAT) == (not kept)
        ctx.close()


@pytest.mark.parametrize("max_plog", [0, 2])
def test_emul_repeat_gate_counted_in_memory(emul, max_plog, monkeypatch):
    """The repeat gate's last resort (where the LDS table gave up the reference completes, :1703-1753): a fragment
    whose duplicated k-mers overflow a pass's table has its distinct k-mers counted in an open-addressing set in memory.
    TGSF_REP_MAX_PLOG forces that at the first / third overflow instead of after 1 024 passes; exact on both sides of -p."""
    monkeypatch.setenv("TGSF_REP_MAX_PLOG", str(max_plog))
    read = _shared_prefix_read()
    c = parity._kmer_repeat_np(read[1], 31)
    for pval, kept in ((c, True), (c + 1, False)):
        p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID], min_q=7.0, min_repeat=pval, kmer=31), [read])
        ctx = capi.Context(p, 0, emul)
        r, f = ctx.submit(*[x if i != 2 else x[:-1].copy() for i, x in enumerate(synth.pack([read]))])
        assert len(f) == 1 and bool(f["flags"][0] & abi.FF_PASS) == kept and bool(f["flags"][0] & abi.FF_REPEAT) == (not kept)
        ctx.close()
    # whole batches through the same path (every k of the keys kernel, 32- and 64-bit keys), against the oracle
    for k in (12, 15, 16, 31):
        reads = parity.repeat_reads(seed=300 + k, n=24)
        p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=8.0, min_repeat=60, kmer=k), reads)
        ctx = capi.Context(p, 0, emul)
        parity.compare_batch(ctx, p, reads)
        ctx.close()


def test_emul_repeat_gate_colliding_hash_values(emul):
    """28 000 distinct repeated 31-mers constructed to share four hash values of the keys kernel: with ONE hash function no
    number of passes separates them (checked once by hand: 2^20 passes, then TGSF_E_CAPACITY); the kernel rotates the hash
    with every restart, and the count is exact on both sides of the gate."""
    parity.colliding_hash_case(emul)


def test_emul_align_windows_beyond_256_bp(emul):
    """Adapters of 257..1280 bp (only reachable with -a; the reference's edlib is multi-block, include/edlib.cpp:182-185) through
    the wide column: edit distance, locations, start and path length as the reference's own edlib reports them."""
    parity.align_windows_random(emul, 600, seed=19, lengths=(257, 300, 511, 640, 1000, 1280), max_window=2600)


def test_emul_batch_with_adapters_beyond_256_bp(emul):
    """Whole batches with a 300-bp and a 700-bp adapter (planted in the middle and at the ends) against the oracle."""
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    a300, a700 = bytes(acgt[rng.integers(0, 4, 300)]), bytes(acgt[rng.integers(0, 4, 700)])
    reads = synth.make_reads(41, 24, "ont", mean_len=6000, zoo=True, pmid=0.6, adapter=a300, err=0.05)
    reads += synth.make_reads(42, 12, "ont", mean_len=9000, zoo=False, pmid=0.8, p5=0.5, adapter=a700, err=0.08)
    for mm in (35, 200):
        p = parity.sized(abi.make_params("ont", adapters=[a300, synth.revcomp(a300), a700, synth.ONT_RAPID], min_q=7.0, mid_match_len=mm, end_match_len=8), reads)
        ctx = capi.Context(p, 0, emul)
        r, f, _ = parity.compare_batch(ctx, p, reads)
        assert (r["flags"] & abi.RF_ADMID).any()
        ctx.close()


@pytest.mark.parametrize("seed", [301348])
def test_emul_fuzz_findings_round3(emul, seed, monkeypatch):
    """Seeds of the round-3 GPU fuzz campaign (TGSF_FUZZ_WIDE=1) that found something.  301348: adapters of 150 and 241 bp at -M 1
    through the position-ordered rescan -- the multi-word scans recorded columns ONE ABOVE their (read, adapter)'s minimum."""
    monkeypatch.setenv("TGSF_FUZZ_WIDE", "1")
    monkeypatch.setenv("TGSF_FUZZ_GATE_P", "0.5")
    from tests import fuzz
    fuzz.run_case(emul, seed, 150)


@pytest.mark.parametrize("seed", [480061])
def test_emul_fuzz_findings_round4(emul, seed, monkeypatch):
    """Seeds of the round-4 GPU fuzz campaign on the filtering middle scan that found something.  480061: a read whose best
    bottom-row value for the 45-bp adapter is k + 1 = 12, in a marked chunk -- k_mid_recheck handed the columns AT k + 1 over
    (its running best starts there) and the read was dropped for a middle adapter it does not hold."""
    monkeypatch.setenv("TGSF_FUZZ_WIDE", "1")
    monkeypatch.setenv("TGSF_FUZZ_GATE_P", "0.2")
    from tests import fuzz
    fuzz.run_case(emul, seed, 150)


@pytest.mark.parametrize("kind,head,tail", [("ont", 79, 0), ("hifi", 7, 8), ("ont", 250, 31), ("hifi", 100, 0), ("ont", 0, 5), ("ont", 99, 1)])
@pytest.mark.parametrize("mode", ["byproduct", None])
def test_emul_clean_tables_as_a_by_product_of_the_raw_pass(emul, kind, head, tail, mode, monkeypatch):
    parity.by_product_run(emul, kind, head, tail, monkeypatch=monkeypatch, mode=mode)


def test_emul_tail_fix_with_tables_longer_than_its_lds_tallies(emul, monkeypatch):
    parity.tail_fix_long_tables(emul, monkeypatch)


@pytest.mark.parametrize("head,tail", [(79, 0), (250, 31), (100, 3), (3, 120)])
def test_emul_by_product_with_quality_bytes_of_128_and_above(emul, head, tail, monkeypatch):
    parity.by_product_high_quality_bytes(emul, head, tail, monkeypatch)


@pytest.mark.parametrize("kind,head,tail", [("ont", 79, 0), ("hifi", 7, 8)])
def test_emul_by_product_when_most_reads_are_kept_as_expected(emul, kind, head, tail, monkeypatch):
    """Few adapters: nearly every read is kept as [head_trim, L - tail_trim) and nothing of it is scanned a second time."""
    parity.by_product_run(emul, kind, head, tail, monkeypatch=monkeypatch, mode=None, p5=0.02, n=120)


@pytest.mark.parametrize("kind,head,tail", [("ont", 79, 0), ("hifi", 7, 8)])
def test_emul_by_product_through_a_pool_overflow(emul, kind, head, tail, monkeypatch):
    """A batch whose candidate pool overflows is run a second time from its inputs: what its first run tallied into the
    clean tables as a by-product is not tallied again, and the second run takes the same decisions."""
    parity.by_product_run(emul, kind, head, tail, monkeypatch=monkeypatch, mode="byproduct", pool_cap=3)


@pytest.mark.parametrize("lengths,n,win", [((1281, 1500, 1800, 1857, 1900, 2048), 60, 4300), ((3000, 5000, 8192), 14, 17000)])
def test_emul_align_windows_beyond_1280_bp(emul, lengths, n, win):
    """Adapters beyond 1 280 bp: where the traceback state of the first location reaches 1 MiB edlib finds its path by
    Hirschberg's divide and conquer (include/edlib.cpp:1191-1210, 1234-1400) and so does alignment_length_w: edit distance,
    locations, start and alignmentLength as the reference's own edlib reports them (the oracle's restatement where
    oracle/_ref is absent)."""
    parity.align_windows_random(emul, n, seed=23, lengths=lengths, max_window=win, plant_whole=True)


def test_emul_batch_with_adapters_beyond_1280_bp(emul):
    """Whole batches with a 2 048-bp and a 3 000-bp adapter (planted in the middle and at the ends) against the oracle."""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    a2k, a3k = bytes(acgt[rng.integers(0, 4, 2048)]), bytes(acgt[rng.integers(0, 4, 3000)])
    reads = []
    for i in range(9):
        L = int(rng.integers(8000, 12000))
        sq = bytearray(acgt[rng.integers(0, 4, L)].tobytes())
        ad = a2k if i % 2 == 0 else a3k
        m = synth.mutate(rng, ad, float(rng.choice([0.0, 0.03, 0.1])))
        if i % 3 == 0:                                  # at the 5' end, after a few bases
            pos = int(rng.integers(0, 40))
        elif i % 3 == 1:                                # in the middle
            pos = int(rng.integers(2700, L - 2700 - len(m)))
        else:                                           # at the 3' end
            pos = L - len(m) - int(rng.integers(0, 40))
        if i < 8:
            sq[pos:pos + len(m)] = m
        reads.append((b"giant%d" % i, bytes(sq), bytes((rng.integers(15, 35, L) + 33).astype(np.uint8))))
    p = parity.sized(abi.make_params("ont", adapters=[a2k, a3k], min_q=7.0, mid_match_len=1200, end_match_len=900, end_len=2600), reads)
    ctx = capi.Context(p, 0, emul)
    r, f, _ = parity.compare_batch(ctx, p, reads)
    assert (r["flags"] & (abi.RF_ADMID | abi.RF_AD5P)).any()
    ctx.close()
