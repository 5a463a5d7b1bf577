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


def _edge_reads(seed=5):
    """Tiny reads (1..8 bases: windows below 5, no middle), reads around the E / 2E+Q boundaries."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = []
    for i, L in enumerate([1, 2, 3, 4, 5, 6, 7, 8, 49, 50, 51, 99, 100, 101, 149, 150, 151, 215, 216, 217, 299, 300, 301,
                           349, 350, 351, 999, 1000, 1001, 6399, 6400, 6401, 12800, 12801]):
        s = acgt[rng.integers(0, 4, L)].tobytes()
        q = (rng.integers(5, 40, L) + 33).astype(np.uint8).tobytes()
        reads.append((b"e%d" % i, s, q))
    return reads


def test_gpu_edge_lengths():
    reads = _edge_reads()
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=7.0,
                                     min_len=100, head_trim=3, tail_trim=2), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads)
    ctx.reset_counters()
    parity.compare_batch(ctx, p, reads, align=1, explicit_lengths=False)
    ctx.close()


def test_gpu_phred64():
    reads = [(n, s, bytes(b + 31 for b in q)) for n, s, q in synth.make_reads(41, 60, "ont", mean_len=4000, zoo=True)]
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=10.0, qtype=64), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


def test_gpu_ultra_long_reads():
    """Config C5 shape: a few reads of 0.3-2 Mb among ordinary ones (hundreds of stats tiles and
    middle segments per read, candidate lists across many lanes)."""
    rng = np.random.default_rng(9)
    reads = synth.make_reads(9, 24, "ont", mean_len=20000, zoo=True, pmid=0.2)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i, L in enumerate([300_000, 1_000_003, 2_000_000]):
        s = bytearray(acgt[rng.integers(0, 4, L)].tobytes())
        for frac in (0.2, 0.5, 0.9):
            p0 = int(L * frac)
            a = synth.mutate(rng, synth.ONT_RAPID if i % 2 else synth.ONT_RAPID_RC, 0.04)
            s[p0:p0 + len(a)] = a
        q = (np.clip(np.rint(rng.normal(14, 4, L)), 1, 50) + 33).astype(np.uint8).tobytes()
        reads.append((b"ul%d" % i, bytes(s), q))
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=10.0), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


@pytest.mark.parametrize("kind,align", [("ont", 16), ("ont", 1), ("hifi", 16)])
def test_gpu_quality_bytes_of_128_and_above(kind, align):
    """Such a byte stands for its value - 256, as in the reference's arithmetic on a signed char (round 5: no longer refused)."""
    reads = parity.high_quality_byte_reads(kind=kind)
    p = parity.sized(abi.make_params(kind, adapters=[synth.ONT_RAPID if kind == "ont" else synth.PACBIO_BLUNT], min_q=7.0, head_trim=13, tail_trim=4), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads, align=align)
    ctx.close()
    p.max_batch_bases = 2 * sum(len(r[1]) for r in reads) + 64 * len(reads) + 4096      # the FASTQ text itself as the batch
    ctx = capi.Context(p, 0)
    parity.compare_batch_in_place(ctx, p, reads)
    ctx.close()


def test_gpu_mean_quality_outside_the_tables_is_reported():
    """A read whose mean of `qual - qType` falls below 0 makes the reference index rawDiffQualReadsBases out of bounds (:1943): refused."""
    reads = synth.make_reads(43, 8, "ont", mean_len=2000)
    n, s, q = reads[3]
    reads[3] = (n, s, bytes([200]) * len(q))
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID]), reads)
    ctx = capi.Context(p, 0)
    seq, qual, off, ln = synth.pack(reads)
    with pytest.raises(capi.TgsfError) as ei:
        ctx.submit(seq, qual, off[:-1].copy(), ln)
    assert ei.value.code == abi.E_DATA
    ctx.close()


def test_gpu_loose_thresholds():
    """-M far below the default: most columns are within k, every lane records ties all the time."""
    reads = synth.make_reads(8, 120, "ont", mean_len=6000, zoo=True, pmid=0.3)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=8.0,
                                     mid_match_len=18, end_match_len=8, mid_sim=0.8, end_sim=0.7), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


def test_gpu_in_place_fastq_text():
    """Both streams read in place from the raw FASTQ text (separate seq / qual offsets, any alignment)."""
    reads = synth.make_reads(12, 300, "ont", mean_len=5000, zoo=True, pmid=0.1)
    p = abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0, head_trim=4)
    p.max_batch_reads = len(reads)
    p.max_batch_bases = 2 * sum(len(r[1]) for r in reads) + 64 * len(reads) + 4096
    p.max_read_len = max(len(r[1]) for r in reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch_in_place(ctx, p, reads)
    ctx.close()


@pytest.mark.parametrize("pval,k", [(300, 11), (40, 9), (2000, 12), (100, 13), (40, 15), (30, 21), (25, 31), (4000, 32)])
def test_gpu_repeat_gate(pval, k):
    """-p/-k: GetKmerCount on the device (LDS bitmap partitions) against the oracle."""
    reads = parity.repeat_reads(n=150)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=10.0,
                                     min_repeat=pval, kmer=k), reads)
    ctx = capi.Context(p, 0)
    res, frags, ctr = parity.compare_batch(ctx, p, reads)
    assert (frags["flags"] & abi.FF_REPEAT).any()
    ctx.close()


@pytest.mark.parametrize("k,lens", [(k, l) for k in (1, 2, 3, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20, 27, 31) for l in ("TINY", "SHORT")]
                         + [(k, "LONG") for k in (3, 10, 11, 12, 13, 15, 16, 31)])
def test_gpu_repeat_gate_exact_counts(k, lens):
    """Reads built to have repeat == T and == T-1 exactly, on the chunk and window seams of k_repeat."""
    parity.repeat_threshold_case(None, k, getattr(parity, "REPEAT_" + lens), max_runs=20)


@pytest.mark.parametrize("k", [13, 14, 20, 31])
@pytest.mark.parametrize("alphabet", [b"AC", b"AT", b"A"])
def test_gpu_repeat_gate_skewed_composition(k, alphabet):
    """Long reads over one or two letters: a pass's table fills up and the fragment starts over with more passes."""
    parity.repeat_threshold_case(None, k, parity.REPEAT_SKEW, max_runs=6, alphabet=alphabet)


@pytest.mark.parametrize("k,plant", [(13, 0), (15, 90), (16, 0), (16, 95), (22, 140), (31, 60)])
def test_gpu_repeat_gate_pass_seams(k, plant):
    """k_repeat_keys around its pass sizes (one pass, two, four), -p on either side of what the first scan flags and of
    the exact count (see test_emul_repeat_gate_pass_seams)."""
    parity.repeat_threshold_case(None, k, parity.REPEAT_SHARE, max_runs=8, share=True, plant=plant, extra_thresholds=(40, 100, 2500))


# The first middle scan of a batch: k_mid_scan1 (TGSF_MID_FLAT=0) and k_mid_flat under stretch schedules from one chunk
# per stretch (every read cut into hundreds of stretches, each with its warm-up) to the default
MID_SCAN_ENVS = [{"TGSF_MID_FLAT": "0"},
                 {"TGSF_FLAT_PMIN": "1", "TGSF_FLAT_PMAX": "1", "TGSF_FLAT_F0": "128"},
                 {"TGSF_FLAT_PMIN": "1", "TGSF_FLAT_PMAX": "8", "TGSF_FLAT_F0": "100"},
                 {"TGSF_FLAT_PMIN": "4", "TGSF_FLAT_PMAX": "64", "TGSF_FLAT_F0": "250"},
                 {"TGSF_FLAT_PMIN": "16", "TGSF_FLAT_PMAX": "256", "TGSF_FLAT_F0": "224"}]


@pytest.mark.parametrize("env", MID_SCAN_ENVS, ids=lambda e: "-".join(f"{k[5:].lower()}{v}" for k, v in e.items()))
def test_gpu_mid_scan_variants(golden_dir, env):
    parity.mid_scan_variants(None, golden_dir, env)


MID_FILTER_ENVS = [{"TGSF_MID_FILTER": "0"}, {"TGSF_MID_FILTER": "1"}, {"TGSF_MID_FILTER": "2"},
                   {"TGSF_MID_FILTER": "2", "TGSF_FLAT_PMIN": "1", "TGSF_FLAT_PMAX": "4", "TGSF_FLAT_F0": "128"},
                   {"TGSF_MID_FILTER": "1", "TGSF_RECHECK_CAP": "5"}]        # (a list of five marks: the rest rechecked where found)


@pytest.mark.parametrize("env", MID_FILTER_ENVS, ids=lambda e: "-".join(f"{k[5:].lower()}{v}" for k, v in e.items()))
def test_gpu_mid_filter(env):
    parity.mid_filter_cases(None, env)


@pytest.mark.parametrize("mode", ["direct", "difference"])
def test_gpu_clean_table_strategy(golden_dir, mode):
    parity.clean_table_strategy(None, mode, golden_dir)


@pytest.mark.parametrize("mode", ["direct", "difference"])
def test_gpu_clean_table_strategy_large(mode, monkeypatch):
    """Thousands of reads: many waves mixing fragments to add with whole reads to take back out."""
    monkeypatch.setenv("TGSF_CLEAN_TABLES", mode)
    reads = synth.make_reads(91, 3000, "hifi", mean_len=7000, zoo=True)
    p = parity.sized(abi.make_params("hifi", adapters=[synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC], min_q=15.0), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


@pytest.mark.parametrize("mode", [None, "direct", "difference"])
def test_gpu_no_qual(mode):
    parity.no_qual_batch(None, mode)


def test_gpu_submit_async():
    parity.async_two_contexts(None)


def test_gpu_align_windows_random():
    parity.align_windows_random(None, 4000)
    parity.align_windows_random(None, 4000, seed=10)


def test_gpu_more_than_64_drop_regions():
    parity.many_regions(None)


def _tie_reads(long_len):
    """Reads whose middle-scan minimum is tied column after column."""
    rng = np.random.default_rng(77)
    q = lambda n: bytes((rng.integers(15, 35, n) + 33).astype(np.uint8))
    reads = synth.make_reads(9, 6, "ont", mean_len=3000, zoo=True, pmid=0.5)
    reads.append((b"polyA", b"A" * long_len, q(long_len)))
    reads.append((b"polyT_ends", b"ACGT" * 100 + b"T" * 9000 + b"GATTACA" * 60, q(400 + 9000 + 420)))
    reads.append((b"ct", b"CT" * 40000, q(80000)))
    reads.append((b"polyA_short", b"A" * 700, q(700)))
    return reads


@pytest.mark.parametrize("adapters,m_mid,long_len", [([b"A" * 50, b"T" * 50], 35, 200_000),
                                                     ([synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC], 1, 200_000),
                                                     ([b"A" * 50, b"T" * 50], 35, 2_000_000)])
def test_gpu_candidate_pool_overflow_is_handled(adapters, m_mid, long_len, monkeypatch, capfd):
    """With the sizing hints a minimal caller passes, a 200-kb (2-Mb) homopolymer read against a homopolymer adapter, or a
    (CT)n read against the PacBio blunt adapter at -M 1, used to end in TGSF_E_CAPACITY; the reference completes them
    (every column at the global minimum is a location, include/edlib.cpp:660-672).  The library re-runs the scan with a pool
    that fits: records and tallies equal the oracle's."""
    monkeypatch.setenv("TGSF_TRACE_POOL", "1")
    reads = _tie_reads(long_len)
    p = parity.sized(abi.make_params("ont", adapters=adapters, min_q=7.0, mid_match_len=m_mid, end_match_len=4), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads)
    err = capfd.readouterr().err
    assert "candidate pool overflow" in err and "(grown)" in err
    ctx.close()


@pytest.mark.parametrize("name", ["ont_zoo", "hifi_zoo", "ont_m1", "huge_adapter", "ont_trim"])
def test_gpu_golden_through_the_overflow_path(golden_dir, name, monkeypatch, capfd):
    monkeypatch.setenv("TGSF_POOL_CAP", "2")
    monkeypatch.setenv("TGSF_TRACE_POOL", "1")
    parity.golden_case(None, golden_dir, name)
    assert "candidate pool overflow" in capfd.readouterr().err


def test_gpu_pool_overflow_device_batches(monkeypatch):
    """tgsf_submit_device: one batch + tgsf_wait takes the fallback (results as without the overflow); so do several
    batches enqueued without a wait between them (the reference completes such inputs, include/edlib.cpp:660-672)."""
    import torch
    reads = synth.make_reads(22, 40, "ont", mean_len=2500, zoo=True, pmid=1.0)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=7.0), reads)
    seq, qual, offsets, lengths = synth.pack(reads)
    ref = capi.Context(p, 0)
    exp_r, exp_f = ref.submit(seq, qual, offsets[:-1].copy(), lengths)
    exp_ctr = ref.counters()
    ref.close()
    monkeypatch.setenv("TGSF_POOL_CAP", "2")
    ctx = capi.Context(p, 0)
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(v).to(dev) for k, v in (("seq", seq), ("qual", qual), ("off", offsets[:-1].astype(np.int64)), ("len", lengths.astype(np.int32)))}
    n, fcap = len(reads), len(exp_f) + 64
    d_reads = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    d_frags = torch.zeros(fcap * 24, dtype=torch.uint8, device=dev)
    d_nf = torch.zeros(4, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)

    def go():
        ctx.submit_device(d["seq"].data_ptr(), d["qual"].data_ptr(), d["off"].data_ptr(), d["len"].data_ptr(), n, seq.size,
                          d_reads.data_ptr(), d_frags.data_ptr(), fcap, d_nf.data_ptr(), st.cuda_stream)
    go()
    st.synchronize()
    ctx.wait()
    st.synchronize()
    got_r = d_reads.cpu().numpy().view(abi.READ_RESULT_DTYPE)
    got_f = d_frags.cpu().numpy().view(abi.FRAGMENT_DTYPE)[:int(d_nf[0].item())]
    assert np.array_equal(got_r, exp_r) and np.array_equal(got_f, exp_f) and np.array_equal(ctx.counters(), exp_ctr)
    ctx.close()
    ctx = capi.Context(p, 0)          # (the first context has grown its pool by now: a new one starts from the forced 2 slots)
    # Three batches enqueued without a wait between them, each into buffers of its own: every one overflows, is left alone
    # by its first run (the fragment count says so to a caller that only synchronises its stream) and is run again by
    # tgsf_wait from its inputs; the raw tallies of the first runs are not added twice.
    outs = [(torch.zeros(n * 32, dtype=torch.uint8, device=dev), torch.zeros(fcap * 24, dtype=torch.uint8, device=dev),
             torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(3)]
    for o_r, o_f, o_n in outs:
        ctx.submit_device(d["seq"].data_ptr(), d["qual"].data_ptr(), d["off"].data_ptr(), d["len"].data_ptr(), n, seq.size,
                          o_r.data_ptr(), o_f.data_ptr(), fcap, o_n.data_ptr(), st.cuda_stream)
    st.synchronize()
    for _, _, o_n in outs:
        assert int(o_n[0].item()) & 0xFFFFFFFF == abi.NFRAGS_NOT_FINAL
    ctx.wait()
    st.synchronize()
    for o_r, o_f, o_n in outs:
        got_r = o_r.cpu().numpy().view(abi.READ_RESULT_DTYPE)
        got_f = o_f.cpu().numpy().view(abi.FRAGMENT_DTYPE)[:int(o_n[0].item())]
        assert np.array_equal(got_r, exp_r) and np.array_equal(got_f, exp_f)
    ref = capi.Context(p, 0)
    for _ in range(3):
        ref.submit(seq, qual, offsets[:-1].copy(), lengths)
    assert np.array_equal(ctx.counters(), ref.counters())
    ref.close()
    ctx.close()


@pytest.mark.parametrize("ads_key", ["lig28", "lig22+28", "lig22+rapid", "mixed", "three"])
@pytest.mark.parametrize("no32", ["0", "1"])
def test_gpu_short_adapters_dword_column(ads_key, no32, monkeypatch):
    """Adapters of at most 32 bp (the reference's ligation-kit library entries, src/TGSFilter.cpp:2974-2977) run the middle scan
    with the one-dword column, two adapters a pass: same locations as the 64-bit column (TGSF_NO_HOT32=1) and as the oracle."""
    from tests.test_emul_parity import ONT_LIGATION_22, ONT_LIGATION_22_RC, ONT_LIGATION_28, ONT_LIGATION_28_RC
    ads = {"lig28": [ONT_LIGATION_28, ONT_LIGATION_28_RC], "lig22+28": [ONT_LIGATION_22, ONT_LIGATION_22_RC, ONT_LIGATION_28, ONT_LIGATION_28_RC],
           "lig22+rapid": [ONT_LIGATION_22, ONT_LIGATION_22_RC, synth.ONT_RAPID, synth.ONT_RAPID_RC], "mixed": [b"ACGTTGCA" * 4, ONT_LIGATION_22],
           "three": [ONT_LIGATION_28, ONT_LIGATION_22_RC, b"ACGTTGCA" * 4]}[ads_key]     # (at most two of <= 32 bp a pass: passes of two and one)
    monkeypatch.setenv("TGSF_NO_HOT32", no32)
    reads = synth.make_reads(31, 400, "ont", mean_len=9000, zoo=True, pmid=0.5, adapter=ads[0], err=0.06)
    p = parity.sized(abi.make_params("ont", adapters=ads, min_q=7.0, mid_match_len=14, end_match_len=4), reads)
    ctx = capi.Context(p, 0)
    r, f, _ = parity.compare_batch(ctx, p, reads)
    assert (r["flags"] & abi.RF_ADMID).any()
    ctx.close()


@pytest.mark.parametrize("k", [31, 15])
def test_gpu_repeat_gate_shared_prefix_fragment(k):
    """Thousands of distinct duplicated k-mers sharing their first 16 (k = 15: 8) bases: the keys kernel starts over with as
    many passes (by a hash of the whole key) as its table needs; the count stays exact (numpy, both sides of the gate)."""
    from tests.test_emul_parity import _shared_prefix_read
    rng = np.random.default_rng(9)
    if k == 31:
        read = _shared_prefix_read()
    else:
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        pre = acgt[rng.integers(0, 4, 8)].tobytes()
        block = b"".join(pre + acgt[rng.integers(0, 4, 7)].tobytes() for _ in range(14000))
        s = block + block
        read = (b"shared_prefix15", s, bytes((rng.integers(15, 35, len(s)) + 33).astype(np.uint8)))
    c = parity._kmer_repeat_np(read[1], k)
    for pval, kept in ((c, True), (c + 1, False)):
        p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID], min_q=7.0, min_repeat=pval, kmer=k), [read])
        ctx = capi.Context(p, 0)
        seq, qual, off, ln = synth.pack([read])
        r, f = ctx.submit(seq, qual, off[:-1].copy(), ln)
        assert len(f) == 1 and bool(f["flags"][0] & abi.FF_PASS) == kept and bool(f["flags"][0] & abi.FF_REPEAT) == (not kept)
        ctx.close()


@pytest.mark.parametrize("k", [15, 31])
@pytest.mark.parametrize("max_plog", [0, 3])
def test_gpu_repeat_gate_counted_in_memory(k, max_plog, monkeypatch):
    """The repeat gate's last resort: a fragment whose duplicated k-mers overflow a pass's LDS table has its distinct
    k-mers counted in an open-addressing set in memory (the reference's unordered_set, src/TGSFilter.cpp:1703-1753) -- forced
    here at the first / fourth overflow instead of after 1 024 passes (TGSF_REP_MAX_PLOG); exact on both sides of -p, and in
    whole batches (several workgroups taking turns at the one table)."""
    monkeypatch.setenv("TGSF_REP_MAX_PLOG", str(max_plog))
    test_gpu_repeat_gate_shared_prefix_fragment(k)
    reads = parity.repeat_reads(seed=400 + k, n=200)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=8.0, min_repeat=60, kmer=k), reads)
    ctx = capi.Context(p, 0)
    parity.compare_batch(ctx, p, reads)
    ctx.close()


def test_gpu_repeat_gate_colliding_hash_values():
    """Distinct repeated 31-mers constructed to share their hash value (see test_emul_repeat_gate_colliding_hash_values)."""
    parity.colliding_hash_case(None)


def test_gpu_align_windows_beyond_256_bp():
    """Adapters of 257..1280 bp through the wide column against the reference's own edlib (where oracle/_ref is present)."""
    parity.align_windows_random(None, 600, seed=19, lengths=(257, 300, 511, 640, 1000, 1280), max_window=2600)


def test_gpu_batch_with_adapters_beyond_256_bp():
    from tests.test_emul_parity import test_emul_batch_with_adapters_beyond_256_bp as same
    same(None)


@pytest.mark.parametrize("seed", [480061])
def test_gpu_fuzz_findings_round4(seed, monkeypatch):
    """See test_emul_fuzz_findings_round4: columns at exactly k + 1 in a chunk the filter marked."""
    monkeypatch.setenv("TGSF_FUZZ_WIDE", "1")
    monkeypatch.setenv("TGSF_FUZZ_GATE_P", "0.2")
    from tests import fuzz
    fuzz.run_case(None, seed, 150)


@pytest.mark.parametrize("kind,head,tail", [("ont", 79, 0), ("hifi", 7, 8), ("ont", 250, 31), ("hifi", 100, 0), ("ont", 0, 5), ("ont", 99, 1)])
@pytest.mark.parametrize("mode", ["byproduct", None])
def test_gpu_clean_tables_as_a_by_product_of_the_raw_pass(kind, head, tail, mode, monkeypatch):
    parity.by_product_run(None, kind, head, tail, monkeypatch=monkeypatch, mode=mode)


def test_gpu_tail_fix_with_tables_longer_than_its_lds_tallies(monkeypatch):
    parity.tail_fix_long_tables(None, monkeypatch)


@pytest.mark.parametrize("head,tail", [(79, 0), (250, 31), (100, 3), (3, 120)])
def test_gpu_by_product_with_quality_bytes_of_128_and_above(head, tail, monkeypatch):
    parity.by_product_high_quality_bytes(None, head, tail, monkeypatch, large=True)


@pytest.mark.parametrize("kind,head,tail", [("ont", 79, 0), ("hifi", 7, 8)])
def test_gpu_by_product_when_most_reads_are_kept_as_expected(kind, head, tail, monkeypatch):
    """Few adapters: nearly every read is kept as [head_trim, L - tail_trim) and nothing of it is scanned a second time."""
    parity.by_product_run(None, kind, head, tail, monkeypatch=monkeypatch, mode=None, p5=0.02, n=120)


@pytest.mark.parametrize("kind,head,tail", [("ont", 79, 0), ("hifi", 7, 8)])
def test_gpu_by_product_through_a_pool_overflow(kind, head, tail, monkeypatch):
    """A batch whose candidate pool overflows is run a second time from its inputs: what its first run tallied into the
    clean tables as a by-product is not tallied again, and the second run takes the same decisions."""
    parity.by_product_run(None, kind, head, tail, monkeypatch=monkeypatch, mode="byproduct", pool_cap=3)


@pytest.mark.parametrize("lengths,n,win", [((1281, 1500, 1800, 1857, 1900, 2048), 60, 4300), ((3000, 5000, 8192), 14, 17000)])
def test_gpu_align_windows_beyond_1280_bp( lengths, n, win):
    """Adapters beyond 1 280 bp: where the traceback state of the first location reaches 1 MiB edlib finds its path by
    Hirschberg's divide and conquer (include/edlib.cpp:1191-1210, 1234-1400) and so does alignment_length_w: edit distance,
    locations, start and alignmentLength as the reference's own edlib reports them (the oracle's restatement where
    oracle/_ref is absent)."""
    parity.align_windows_random(None, n, seed=23, lengths=lengths, max_window=win, plant_whole=True)


def test_gpu_batch_with_adapters_beyond_1280_bp():
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
    ctx = capi.Context(p, 0)
    r, f, _ = parity.compare_batch(ctx, p, reads)
    assert (r["flags"] & (abi.RF_ADMID | abi.RF_AD5P)).any()
    ctx.close()
