"""Shared parity checks: a libtgsf backend (HIP on the GPU box, the serial emulation of the same
kernels on GPU-less boxes) against the oracle and the golden vectors."""
from __future__ import annotations

import json
import os

import numpy as np

from oracle import orc
from tests import hostmodel
from tgsfilter_amd import abi, capi, synth


def sized(p: abi.Params, reads, extra=1):
    n = len(reads)
    p.max_batch_reads = n * extra
    p.max_batch_bases = sum(len(r[1]) for r in reads) * extra + 64
    p.max_read_len = max(len(r[1]) for r in reads)
    return p


def compare_batch(ctx: capi.Context, p: abi.Params, reads, *, align=16, explicit_lengths=True):
    """Run one batch through the backend and the oracle; everything must be identical."""
    seq, qual, offsets, lengths = synth.pack(reads, align=align)
    if explicit_lengths:
        got_r, got_f = ctx.submit(seq, qual, offsets[:-1].copy() if align > 1 else offsets, lengths)
    else:
        got_r, got_f = ctx.submit(seq, qual, offsets, None)
    ctr = ctx.counters()
    exp_ctr = np.zeros(ctx.ctr_words, dtype=np.uint64)
    exp_r, exp_f, exp_ctr = orc.filter_batch(p, seq, qual, offsets, lengths if explicit_lengths else None,
                                             n_bins=ctx.n_bins, ctr=exp_ctr)
    for name in ("sum_q", "flags", "n_frags", "frag_begin", "trimmed"):
        bad = np.nonzero(got_r[name] != exp_r[name])[0]
        assert bad.size == 0, f"read field {name} differs at reads {bad[:8]}: got {got_r[name][bad[:8]]} exp {exp_r[name][bad[:8]]}"
    assert len(got_f) == len(exp_f), (len(got_f), len(exp_f))
    for name in ("sum_q", "read", "start", "len", "flags"):
        bad = np.nonzero(got_f[name] != exp_f[name])[0]
        assert bad.size == 0, f"fragment field {name} differs at {bad[:8]}"
    bad = np.nonzero(ctr != exp_ctr)[0]
    assert bad.size == 0, f"tally words differ at {bad[:12]}: got {ctr[bad[:12]]} exp {exp_ctr[bad[:12]]}"
    return got_r, got_f, ctr


def high_quality_byte_reads(seed=61, n=40, kind="ont"):
    """Reads with quality bytes of 128 and above sprinkled in: the reference subtracts qType from a (signed) char
    (src/TGSFilter.cpp:1455-1457, :1508), so such a byte stands for its value - 256.  Every case keeps the read's
    mean in [0, 256) (outside it the reference indexes rawDiffQualReadsBases out of bounds, :1943)."""
    reads = synth.make_reads(seed, n, kind, mean_len=3000, zoo=True, pmid=0.05)
    rng = np.random.default_rng(seed)
    out = []
    for i, (name, sq, q) in enumerate(reads):
        q = bytearray(q)
        L = len(q)
        if i % 4 == 0 and L > 400:                     # a few, anywhere (tile seams, read ends, every alignment)
            for pos in list(rng.integers(0, L, max(1, L // 300))) + [0, L - 1, 99, 100, min(L - 1, 6399), min(L - 1, 6400)]:
                q[int(pos)] = int(rng.integers(128, 256))
            for j in range(L):                           # ... beside high ordinary qualities: the mean stays positive
                if q[j] < 128:
                    q[j] = max(q[j], 33 + 35)
        elif i % 4 == 1 and L > 400:                   # a run of them over a whole 100-base bin and across a dword
            a = int(rng.integers(0, L - 210))
            for j in range(a, a + 205):
                q[j] = 255 if j % 3 else 128
            for j in range(L):
                if q[j] < 128:
                    q[j] = 126
        out.append((name, sq, bytes(q)))
    return out


def by_product_run(lib_path, kind, head, tail, *, monkeypatch, mode="byproduct", pool_cap=None, p5=None, n=48, batches=3):
    """The clean tables as a by-product of the raw pass (round 5, k_stats<raw, BP>): reads expected to be kept as
    [head_trim, L - tail_trim) are tallied into the clean tables while the raw pass holds them; the others are corrected
    afterwards.  Several batches through ONE context (the device decides per batch whether the next one speculates), every
    record, fragment and tally word against the oracle -- whatever the guesses were, the results may not depend on them."""
    from oracle import orc
    monkeypatch.setenv("TGSF_CLEAN_TABLES", mode) if mode else monkeypatch.delenv("TGSF_CLEAN_TABLES", raising=False)
    if pool_cap:
        monkeypatch.setenv("TGSF_POOL_CAP", str(pool_cap))
    ad = [synth.ONT_RAPID, synth.ONT_RAPID_RC] if kind == "ont" else [synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC]
    sets = []
    for b in range(batches):
        reads = synth.make_reads(300 + 7 * b + head, n, kind, mean_len=3000 + 2500 * b, zoo=(b == 1), pmid=0.05 if b else 0.0,
                                 p5=p5 if p5 is not None else (0.8 if kind == "ont" else 0.1))
        if b == 2:                                     # lengths on the seams of bins, tiles and the trims themselves
            rng = np.random.default_rng(b)
            acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
            for L in (head + tail + 999, head + tail + 1000, head + tail + 1001, 6400, 6400 + head, 6400 + head + 1, 6399 + head + tail, 12800 + head, 6500, 1100 + head):
                reads.append((b"seam%d" % L, bytes(acgt[rng.integers(0, 4, L)]), bytes((rng.integers(25, 40, L) + 33).astype(np.uint8))))
        sets.append(reads)
    p = sized(abi.make_params(kind, adapters=ad, min_q=9.0 if kind == "ont" else 20.0, head_trim=head, tail_trim=tail), [r for rs in sets for r in rs])
    ctx = capi.Context(p, 0, lib_path)
    exp = np.zeros(ctx.ctr_words, dtype=np.uint64)
    try:
        for k, reads in enumerate(sets + sets[:1]):
            seq, qual, offsets, lengths = synth.pack(reads, align=16 if k % 2 == 0 else 1)
            got_r, got_f = ctx.submit(seq, qual, offsets[:-1].copy() if k % 2 == 0 else offsets, lengths)
            exp_r, exp_f, exp = orc.filter_batch(p, seq, qual, offsets, lengths, n_bins=ctx.n_bins, ctr=exp)
            for name in ("sum_q", "flags", "n_frags", "frag_begin", "trimmed"):
                assert np.array_equal(got_r[name], exp_r[name]), (k, name)
            assert len(got_f) == len(exp_f)
            for name in ("sum_q", "read", "start", "len", "flags"):
                bad = np.nonzero(got_f[name] != exp_f[name])[0]
                assert bad.size == 0, (k, name, bad[:8], got_f[name][bad[:8]], exp_f[name][bad[:8]])
            ctr = ctx.counters()
            bad = np.nonzero(ctr != exp)[0]
            assert bad.size == 0, f"batch {k}: tally words differ at {bad[:12]}: got {ctr[bad[:12]]} exp {exp[bad[:12]]}"
    finally:
        ctx.close()


def by_product_high_quality_bytes(lib_path, head, tail, monkeypatch, large=False):
    """The split-bin by-product (round 6) with quality bytes of 128 and above in head pieces, tail pieces and behind the
    speculated fragment: 256 per such byte comes back out of the raw row AND of the clean row its piece belongs to."""
    monkeypatch.setenv("TGSF_CLEAN_TABLES", "byproduct")
    reads = high_quality_byte_reads(seed=62 + head, n=600 if large else 40, kind="ont")
    for align in (16, 1):
        p = sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=7.0, head_trim=head, tail_trim=tail), reads)
        ctx = capi.Context(p, 0, lib_path)
        try:
            compare_batch(ctx, p, reads, align=align)
        finally:
            ctx.close()


def tail_fix_long_tables(lib_path, monkeypatch):
    """k_tail_fix's other way (round 6): a context whose per-100-bp tables are longer than its LDS tallies hold (reads of more
    than 102 400 bp may come) adds the bytes behind a speculated fragment straight to the tables in memory."""
    monkeypatch.setenv("TGSF_CLEAN_TABLES", "byproduct")
    reads = synth.make_reads(77, 60, "ont", mean_len=4000, zoo=True, pmid=0.05, p5=0.3)
    p = sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=7.0, head_trim=79, tail_trim=13), reads)
    p.max_read_len = 250_000                                          # 2 501 rows
    ctx = capi.Context(p, 0, lib_path)
    try:
        assert ctx.n_bins > 1024
        compare_batch(ctx, p, reads, align=16)
    finally:
        ctx.close()


def golden_case(lib_path, golden_dir, name):
    case = hostmodel.GoldenCase(golden_dir, name)
    p = sized(case.params(), case.reads)
    ctx = capi.Context(p, 0, lib_path)
    try:
        res, frags, ctr = compare_batch(ctx, p, case.reads)
        if name != "qc_only":
            assert hostmodel.format_fastq(case.reads, res, frags) == case.ref_out
            drop = ctr[abi.CTR_DROPINFO:abi.CTR_DROPINFO + 17]
            for i, v in enumerate(case.info["drop"]):
                if v is not None:
                    assert int(drop[i]) == v, (i, int(drop[i]), v)
    finally:
        ctx.close()


def edlib_vectors(lib_path, golden_dir):
    vec = json.load(open(os.path.join(golden_dir, "edlib_vectors.json")))
    adapters = sorted({v["q"] for v in vec})
    p = abi.make_params("ont", adapters=[a.encode() for a in adapters], max_batch_bases=1 << 16,
                        max_batch_reads=64, max_read_len=4096)
    ctx = capi.Context(p, 0, lib_path)
    try:
        buf, off, ln, aid, ks = bytearray(), [], [], [], []
        for v in vec:
            off.append(len(buf)); ln.append(len(v["t"])); aid.append(adapters.index(v["q"])); ks.append(v["k"])
            buf += v["t"].encode()
        res, ends = ctx.align_windows(bytes(buf), off, ln, aid, ks)
        for i, v in enumerate(vec):
            exp = (v["ed"], v["n"], v["alen"], v["starts"][0] if v["n"] else -1,
                   v["ends"][0] if v["n"] else -1, v["ends"][-1] if v["n"] else -1)
            got = (int(res[i, 0]), int(res[i, 1]), int(res[i, 2]), int(res[i, 3]), int(ends[i, 0]), int(ends[i, 1]))
            assert got == exp, (i, v["q"], v["t"], v["k"], got, exp)
    finally:
        ctx.close()


def fastq_text_layout(reads):
    """The raw FASTQ text of `reads` plus the in-place index a caller would build over it."""
    buf = bytearray()
    off, qoff, ln = [], [], []
    for name, s, q in reads:
        buf += b"@" + name + b"\n"
        off.append(len(buf)); buf += s + b"\n+\n"
        qoff.append(len(buf)); buf += q + b"\n"
        ln.append(len(s))
    text = np.frombuffer(bytes(buf) + b"\0" * 64, dtype=np.uint8)
    return text, np.array(off, np.uint64), np.array(qoff, np.uint64), np.array(ln, np.uint32)


def compare_batch_in_place(ctx: capi.Context, p: abi.Params, reads):
    """seq == qual == the FASTQ text itself, two offset arrays: must equal the oracle on the same layout
    (and therefore the packed layout, which the oracle is indifferent to)."""
    text, off, qoff, ln = fastq_text_layout(reads)
    got_r, got_f = ctx.submit(text, text, off, ln, qual_offsets=qoff)
    ctr = ctx.counters()
    exp_r, exp_f, exp_ctr = orc.filter_batch(p, text, text, off, ln, n_bins=ctx.n_bins, qual_offsets=qoff)
    assert np.array_equal(got_r, exp_r)
    assert np.array_equal(got_f, exp_f)
    bad = np.nonzero(ctr != exp_ctr)[0]
    assert bad.size == 0, f"tally words differ at {bad[:12]}: got {ctr[bad[:12]]} exp {exp_ctr[bad[:12]]}"
    seq, qual, offsets, lengths = synth.pack(reads)
    pk_r, pk_f, pk_ctr = orc.filter_batch(p, seq, qual, offsets, lengths, n_bins=ctx.n_bins)
    assert np.array_equal(pk_ctr, exp_ctr) and np.array_equal(pk_f, exp_f)


def repeat_reads(seed=51, n=60):
    """Random reads plus low-complexity (tandem repeat) reads: exercises both sides of the -p gate."""
    reads = synth.make_reads(seed, n, "ont", mean_len=3000, zoo=True)
    rng = np.random.default_rng(seed)
    for i in range(12):
        unit = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(rng.integers(2, 60)))])
        L = int(rng.integers(1500, 9000))
        s = bytearray((unit * (L // len(unit) + 1))[:L])
        for pos in rng.integers(0, L, L // 50):
            s[int(pos)] = int(np.frombuffer(b"ACGTNa", dtype=np.uint8)[rng.integers(0, 6)])
        q = bytes((rng.integers(12, 30, L) + 33).astype(np.uint8))
        reads.append((b"rep%d" % i, bytes(s), q))
    return reads



def _kmer_repeat_np(seq: bytes, k: int) -> int:
    """GetKmerCount (src/TGSFilter.cpp:1703-1753) for k < 32 in numpy: #k-mers - #distinct k-mers."""
    a = np.frombuffer(seq, dtype=np.uint8)
    total = a.size - k + 1
    if total <= 0:
        return 0
    code = np.zeros(256, dtype=np.uint64)
    code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
    c = code[a]
    km = np.zeros(total, dtype=np.uint64)
    for j in range(k):
        km = (km << np.uint64(2)) | c[j:j + total]
    return int(total - np.unique(km).size)


REPEAT_TINY = [100, 101, 111, 112, 113, 127, 128, 129, 130, 143, 144, 145]
REPEAT_SHORT = [500, 1000, 1023, 1024, 1025, 4097, 16384, 20000]
_W = 6 * 1024 * 16      # bases of k_repeat's window (one 16-base chunk is shared between consecutive windows)
REPEAT_SKEW = [400000, 150000, 30000]
REPEAT_LONG = [_W - 40, _W - 17, _W - 16, _W - 15, _W - 1, _W, _W + 1, _W + 15, _W + 16, _W + 17,
               2 * _W - 33, 2 * _W - 16, 2 * _W - 15, 2 * _W + 3, 3 * _W - 20, 250000]


_S = 114688             # k-mers of a single pass of k_repeat_keys (kRepShare)
REPEAT_SHARE = [50000, _S - 1, _S, _S + 1, 2 * _S, 2 * _S + 1, 300000]     # + k - 1 bases each: see repeat_threshold_case(..., share=True)


def repeat_threshold_case(lib_path, k, lens, max_runs=64, alphabet=b"ACGT", share=False, plant=0, extra_thresholds=()):
    """The gate at each read's own count: with -p c the read whose repeat count is c passes, with -p c+1 it is
    dropped -- a miscount by one k-mer in either direction shows.  Lengths sit on the kernel's seams (16-base
    chunks, the window of k_repeat, two and three windows); stray N / lower-case bytes; one low-complexity read;
    a two-letter alphabet crowds the k-mers into a few passes (k_repeat_keys then starts over with more of them);
    the reads start at any alignment (runs alternate between the packed layout and the FASTQ text in place).
    share: the lengths are numbers of k-mers around the passes of k_repeat_keys (one pass, two, four), the reads random
    but for `plant` k-mers copied from elsewhere in the read (so that the count sits near a -p that the flagged
    occurrences of the first scan alone cannot decide); extra_thresholds: -p values besides every read's own count."""
    rng = np.random.default_rng(100 + k)
    reads, counts = [], []
    for i, L in enumerate(lens):
        if share:
            L += k - 1
        body = np.frombuffer(alphabet, dtype=np.uint8)[rng.integers(0, len(alphabet), L)].copy()
        if i == 1 and not share:
            unit = body[:int(rng.integers(3, 40))]
            body = np.tile(unit, L // unit.size + 1)[:L].copy()
        if plant and i != 0:
            n = plant + i + k - 1
            src, dst = int(rng.integers(0, L // 2 - n)), int(rng.integers(L // 2, L - n))
            body[dst:dst + n] = body[src:src + n]
        for pos in rng.integers(0, L, 0 if share else max(1, L // 400)):
            body[int(pos)] = int(np.frombuffer(b"Nacgt", dtype=np.uint8)[rng.integers(0, 5)])
        q = bytes((rng.integers(12, 30, L) + 33).astype(np.uint8))
        reads.append((b"r%d_%s" % (L, b"x" * int(rng.integers(0, 16))), body.tobytes(), q))
        counts.append(_kmer_repeat_np(body.tobytes(), k))
    counts = np.array(counts)
    thresholds = sorted({int(c) + d for c in counts for d in (0, 1) if int(c) + d > 0})
    if len(thresholds) > max_runs:
        thresholds = [thresholds[i] for i in sorted(rng.choice(len(thresholds), max_runs, replace=False))]
    thresholds = sorted(set(thresholds) | set(extra_thresholds))
    for run, T in enumerate(thresholds):
        p = abi.make_params("ont", adapters=[], min_q=5.0, min_len=100, min_repeat=T, kmer=k)
        p.max_batch_reads = len(reads)
        p.max_batch_bases = 2 * sum(len(r[1]) for r in reads) + 64 * len(reads) + 4096
        p.max_read_len = max(len(r[1]) for r in reads)
        ctx = capi.Context(p, 0, lib_path)
        if run & 1:
            compare_batch_in_place(ctx, p, reads)
        else:
            res, frags, ctr = compare_batch(ctx, p, reads, align=1 if run & 2 else 16)
            dropped = (frags["flags"] & abi.FF_REPEAT) != 0
            assert np.array_equal(dropped, counts < T), (T, counts, dropped)
        ctx.close()


def colliding_hash_read(n_special=28000, c=0x5A17C3E8, seed=3):
    """n_special 31-mers, one after the other, built so that tail = (c ^ rotl13(head)) & ~3 (head = the first 16 bases, random;
    tail = the other 15): they all share one of four 32-bit hash values of k_repeat_keys' first attempt -- 7 000 distinct keys
    to a value, more than a pass's table holds, and no number of passes by that hash separates them.  The read is the chain
    twice over, so that every one of them is a repeated key (and has to be compared in full)."""
    rng = np.random.default_rng(seed)
    heads = rng.integers(0, 1 << 32, n_special, dtype=np.uint64)
    rot = ((heads << np.uint64(13)) | (heads >> np.uint64(19))) & np.uint64(0xFFFFFFFF)
    tails = (np.uint64(c) ^ rot) & np.uint64(0xFFFFFFFC)
    codes = np.empty((n_special, 31), dtype=np.uint8)
    for j in range(16):
        codes[:, j] = (heads >> np.uint64(30 - 2 * j)) & np.uint64(3)
    for j in range(15):
        codes[:, 16 + j] = (tails >> np.uint64(30 - 2 * j)) & np.uint64(3)
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[codes.reshape(-1)].tobytes()
    s = s + s
    q = bytes((rng.integers(15, 35, len(s)) + 33).astype(np.uint8))
    return (b"colliding", s, q)


def colliding_hash_case(lib_path):
    """The gate at the read's own count (numpy) and one above it, k = 31, on colliding_hash_read()."""
    read = colliding_hash_read()
    c = _kmer_repeat_np(read[1], 31)
    assert c > 800_000
    for pval, kept in ((c, True), (c + 1, False)):
        p = sized(abi.make_params("ont", adapters=[synth.ONT_RAPID], min_q=7.0, min_repeat=pval, kmer=31), [read])
        ctx = capi.Context(p, 0, lib_path)
        seq, qual, off, ln = synth.pack([read])
        r, f = ctx.submit(seq, qual, off[:-1].copy(), ln)
        assert len(f) == 1 and bool(f["flags"][0] & abi.FF_PASS) == kept and bool(f["flags"][0] & abi.FF_REPEAT) == (not kept)
        ctx.close()


def mid_filter_cases(lib_path, env):
    """k_mid_flat as a filter (adapters of 33..64 bp within at most kSuffixMaxK = 12 differences: the last 32 rows in the
    dword column, marked chunks rechecked by k_mid_recheck) under `env` (TGSF_MID_FILTER: 0 off, 1 / 2 the test stride):
    every -M from an exact match to the first value past the filter's range, the 35/36-bp library adapters with their
    homopolymer tails against homopolymer reads (every chunk marked: k_mid_recheck's list outgrows its LDS, the candidate
    pool overflows and the batch is run again), adapters planted across chunk and read boundaries and in a window's last,
    partial chunk, mixed passes (filtered, unfiltered and one-dword adapters in one context)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        pb = [synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC]
        for mml in (45, 44, 40, 35, 34, 33, 32):                             # k = 1, 2, 6, 11, 12, 13 (unfiltered), 14
            reads = synth.make_reads(900 + mml, 48, "hifi", mean_len=2500, zoo=True, pmid=0.5, err=0.5 * (46 - mml) / 45.0)
            p = sized(abi.make_params("hifi", adapters=pb, mid_match_len=mml, min_len=200), reads)
            ctx = capi.Context(p, 0, lib_path)
            compare_batch(ctx, p, reads, align=1, explicit_lengths=False)
            ctx.close()
        # the kit adapters with homopolymer tails, and reads that are one long homopolymer / dinucleotide repeat
        tails = [b"AAAAAAAAAAAAAAAAAATTAACGGAGGAGGAGGA", b"TCCTCCTCCTCCGTTAATTTTTTTTTTTTTTTTTT",
                 b"TTTTTTTTCCTGTACTTCGTTCAGTTACGTATTGCT", b"AGCAATACGTAACTGAACGAAGTACAGGAAAAAAAA"]
        rng = np.random.default_rng(77)
        reads = synth.make_reads(950, 24, "hifi", mean_len=3000, zoo=True, pmid=0.6, adapter=tails[1])
        for i, (base, L) in enumerate(((b"T", 70000), (b"A", 5000), (b"TC", 9000), (b"T", 331), (b"GGA", 6000))):
            seq = (base * (L // len(base) + 1))[:L]
            reads.insert(3 * i + 1, (b"mono%d" % i, seq, bytes(rng.integers(60, 70, L, dtype=np.uint8))))
        for mml in (35, 30, 24):
            p = sized(abi.make_params("hifi", adapters=tails, mid_match_len=mml, min_len=100), reads)
            ctx = capi.Context(p, 0, lib_path)
            compare_batch(ctx, p, reads)
            ctx.close()
        # an adapter ending in each of the 16 columns of a chunk, in the first / last chunk of a window and across the
        # boundary of two reads' windows (E = 0: the window is the read), with up to k differences
        rng = np.random.default_rng(78)
        reads = []
        for i in range(64):
            L = int(rng.integers(300, 700))
            seq = bytearray(synth._ACGT[rng.integers(0, 4, L)].tobytes())
            a = synth.mutate(rng, pb[i & 1], 0.02 * (i % 6))
            at = (0, L - len(a), L - len(a) - 1 - i % 16, 16 * (i % 9) + i % 16, L // 2 + i % 16)[i % 5]
            at = max(0, min(at, L - len(a)))
            seq[at:at + len(a)] = a
            reads.append((b"edge%d" % i, bytes(seq), bytes(rng.integers(60, 70, L, dtype=np.uint8))))
        for e_len in (0, 7, 150):
            p = sized(abi.make_params("hifi", adapters=pb, end_len=e_len, min_len=50), reads)
            ctx = capi.Context(p, 0, lib_path)
            compare_batch(ctx, p, reads, align=1, explicit_lengths=False)
            ctx.close()
        # reads whose ONLY near-match has k - 1, k, k + 1, k + 2 substitutions or inserted bases (k = 11: spread over the adapter, over its
        # first 13 rows, over its last 32): a column at k + 1 sits in a marked chunk and must not become a candidate (with inserted bases
        # every base of the adapter still matches: such a candidate would pass the match-length gate)
        rng = np.random.default_rng(79)
        reads = []
        comp = {65: 67, 67: 65, 71: 84, 84: 71}
        for i in range(96):
            L = int(rng.integers(400, 900))
            seq = bytearray(synth._ACGT[rng.integers(0, 4, L)].tobytes())
            a = bytearray(pb[i & 1])
            nsub = 10 + (i >> 1) % 4
            where = (np.arange(len(a)), np.arange(13), np.arange(13, len(a)))[(i >> 3) % 3]
            if i & 4:                        # differences that keep every base of the adapter matched: bases inserted into the text
                for x in sorted(rng.choice(where[1:], size=min(nsub, len(where) - 1), replace=False), reverse=True):
                    a[x:x] = bytes([comp[a[x]]])
            else:
                for x in rng.choice(where, size=min(nsub, len(where)), replace=False):
                    a[x] = comp[a[x]]
            at = int(rng.integers(0, L - len(a)))
            seq[at:at + len(a)] = a
            reads.append((b"near%d" % i, bytes(seq), bytes(rng.integers(60, 70, L, dtype=np.uint8))))
        p = sized(abi.make_params("hifi", adapters=pb, end_len=0, min_len=50), reads)
        ctx = capi.Context(p, 0, lib_path)
        compare_batch(ctx, p, reads, align=1, explicit_lengths=False)
        ctx.close()
        # filtered, unfiltered and one-dword adapters in one context, in an order that alternates the classes
        mixed = [synth.PACBIO_BLUNT, synth.ONT_RAPID, synth.PACBIO_BLUNT_RC, synth.ONT_RAPID[:28], tails[2], tails[3],
                 b"CTTGCGGGCGGCGGACTCTCCTCTGAAGATAGAGCGACAGGCAAG", b"CTTGCCTGTCGCTCTATCTTCAGAGGAGAGTCCGCCGCCCGCAAG", synth.ONT_RAPID_RC]
        reads = synth.make_reads(960, 40, "hifi", mean_len=3000, zoo=True, pmid=0.5)
        reads += synth.make_reads(961, 20, "ont", mean_len=3000, zoo=True, pmid=0.5)
        reads += synth.make_reads(962, 20, "hifi", mean_len=3000, zoo=True, pmid=0.5, adapter=mixed[6])
        p = sized(abi.make_params("hifi", adapters=mixed, min_len=200), reads)
        ctx = capi.Context(p, 0, lib_path)
        compare_batch(ctx, p, reads)
        ctx.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def mid_scan_variants(lib_path, golden_dir, env):
    """The first middle scan of a batch under `env`: TGSF_MID_FLAT=0 (k_mid_scan1, one 1 024-column block per lane) or
    k_mid_flat with the given stretch schedule (TGSF_FLAT_PMIN / _PMAX / _F0: stretches down to ONE chunk, so that every
    read is cut into many stretches with their warm-ups, stretches run across read boundaries and phases end in partial
    groups).  Goldens with middle hits and splits, random batches with planted middle adapters, unpadded CSR (windows
    at every alignment), -E below 15 (the last chunk of a window fetched byte by byte), 1 to 4 adapters per pass and
    the one-dword column."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for name in ("ont_zoo", "hifi_zoo", "ont_discard"):
            golden_case(lib_path, golden_dir, name)
        ads = [synth.ONT_RAPID, synth.ONT_RAPID_RC]
        reads = synth.make_reads(801, 70, "ont", mean_len=5000, zoo=True, pmid=0.3)
        p = sized(abi.make_params("ont", adapters=ads, min_q=8.0), reads)
        ctx = capi.Context(p, 0, lib_path)
        compare_batch(ctx, p, reads)
        ctx.close()
        ctx = capi.Context(p, 0, lib_path)
        compare_batch(ctx, p, reads, align=1, explicit_lengths=False)       # unpadded CSR: windows at every alignment
        ctx.close()
        for e_len in (0, 3, 14, 15, 16, 40):                                 # -E: bytes of the read behind the window
            reads = synth.make_reads(802 + e_len, 40, "ont", mean_len=1500, zoo=True, pmid=0.4)
            p = sized(abi.make_params("ont", adapters=ads, min_q=8.0, end_len=e_len, min_len=100), reads)
            ctx = capi.Context(p, 0, lib_path)
            compare_batch(ctx, p, reads, align=1, explicit_lengths=False)
            ctx.close()
        # three and four adapters in one pass (33..64 bp), and a pair of at most 32 bp (the one-dword column)
        for k, adset in enumerate(([synth.ONT_RAPID, synth.ONT_RAPID_RC, synth.PACBIO_BLUNT],
                                   [synth.ONT_RAPID, synth.ONT_RAPID_RC, synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC],
                                   [synth.ONT_RAPID[:28], synth.ONT_RAPID_RC[-28:]])):
            reads = synth.make_reads(820 + k, 40, "ont", mean_len=3000, zoo=True, pmid=0.3)
            p = sized(abi.make_params("ont", adapters=adset, min_q=8.0, mid_match_len=24 if k == 2 else 35), reads)
            ctx = capi.Context(p, 0, lib_path)
            compare_batch(ctx, p, reads)
            ctx.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def clean_table_strategy(lib_path, mode, golden_dir):
    """Both ways of tallying the clean bin tables (TGSF_CLEAN_TABLES=direct|difference, see k_clean_plan)
    must give the oracle's tallies: trimmed / split / dropped / low-quality / repeat-dropped reads, -F,
    in-place text, and two batches through one context (the per-batch raw table is reused)."""
    old = os.environ.get("TGSF_CLEAN_TABLES")
    os.environ["TGSF_CLEAN_TABLES"] = mode
    try:
        for name in ("ont_zoo", "hifi_zoo", "ont_discard"):
            golden_case(lib_path, golden_dir, name)
        ads = [synth.ONT_RAPID, synth.ONT_RAPID_RC]
        # repeat gate + quality gates
        reads = repeat_reads(seed=77, n=40)
        p = sized(abi.make_params("ont", adapters=ads, min_q=10.0, min_repeat=300, kmer=11), reads)
        ctx = capi.Context(p, 0, lib_path)
        compare_batch(ctx, p, reads)
        ctx.close()
        # -F: nothing is trimmed, every read is kept whole
        reads = synth.make_reads(31, 30, "hifi", mean_len=9000, zoo=True)
        p = sized(abi.make_params("hifi", adapters=[synth.PACBIO_BLUNT], filter=0), reads)
        ctx = capi.Context(p, 0, lib_path)
        compare_batch(ctx, p, reads)
        ctx.close()
        # in-place text
        reads = synth.make_reads(32, 30, "ont", mean_len=3000, zoo=True, pmid=0.1)
        p = abi.make_params("ont", adapters=ads, min_q=9.0, head_trim=4)
        p.max_batch_reads = len(reads)
        p.max_batch_bases = 2 * sum(len(r[1]) for r in reads) + 64 * len(reads) + 4096
        p.max_read_len = max(len(r[1]) for r in reads)
        ctx = capi.Context(p, 0, lib_path)
        compare_batch_in_place(ctx, p, reads)
        ctx.close()
        # two batches, one context: tallies accumulate
        r1 = synth.make_reads(33, 24, "hifi", mean_len=12000, zoo=True)
        r2 = synth.make_reads(34, 30, "hifi", mean_len=4000, zoo=True)
        p = sized(abi.make_params("hifi", adapters=[synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC], min_q=15.0), r1 + r2)
        ctx = capi.Context(p, 0, lib_path)
        exp = np.zeros(ctx.ctr_words, dtype=np.uint64)
        for rd in (r1, r2):
            seq, qual, offsets, lengths = synth.pack(rd)
            ctx.submit(seq, qual, offsets[:-1].copy(), lengths)
            orc.filter_batch(p, seq, qual, offsets, lengths, n_bins=ctx.n_bins, ctr=exp)
        got = ctx.counters()
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, f"tally words differ at {bad[:12]}"
        ctx.close()
    finally:
        if old is None:
            os.environ.pop("TGSF_CLEAN_TABLES", None)
        else:
            os.environ["TGSF_CLEAN_TABLES"] = old


def no_qual_batch(lib_path, mode=None):
    """Records without qualities (FASTA input, tgsf_params.no_qual): count-only tallies incl. the reference's
    lower-case quirks at the read ends, no quality gate, adapter trimming as usual."""
    old = os.environ.get("TGSF_CLEAN_TABLES")
    if mode:
        os.environ["TGSF_CLEAN_TABLES"] = mode
    try:
        rng = np.random.default_rng(5)
        reads = []
        for i, (name, s, q) in enumerate(synth.make_reads(41, 60, "ont", mean_len=3000, zoo=True, pmid=0.1)):
            b = bytearray(s)
            if i % 2 == 0:                                 # soft-masked stretches over both ends and the middle
                for lo, hi in ((0, 40), (len(b) - 40, len(b)), (len(b) // 2, len(b) // 2 + 60)):
                    for k in range(max(lo, 0), min(hi, len(b))):
                        if rng.random() < 0.7:
                            b[k] = b[k] | 0x20 if chr(b[k]).isalpha() else b[k]
            reads.append((name, bytes(b), q))
        p = sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=10.0, head_trim=3,
                                  no_qual=True), reads)
        ctx = capi.Context(p, 0, lib_path)
        res, frags, ctr = compare_batch(ctx, p, reads)
        assert not (res["flags"] & abi.RF_LOWQ).any() and (res["sum_q"] == 0).all() and (frags["sum_q"] == 0).all()
        assert ctr[abi.CTR_RAW_DIFFQ:abi.CTR_RAW_DIFFQ + 512].sum() == 0
        ctx.close()
    finally:
        if old is None:
            os.environ.pop("TGSF_CLEAN_TABLES", None)
        else:
            os.environ["TGSF_CLEAN_TABLES"] = old


def async_two_contexts(lib_path):
    """tgsf_submit_async on two contexts, then tgsf_wait on both: each batch equals the oracle's result; a second
    submit on a context with a batch pending is refused."""
    ads = [synth.ONT_RAPID, synth.ONT_RAPID_RC]
    r1 = synth.make_reads(61, 40, "ont", mean_len=3000, zoo=True, pmid=0.1)
    r2 = synth.make_reads(62, 50, "ont", mean_len=2500, zoo=True, pmid=0.1)
    p = sized(abi.make_params("ont", adapters=ads, min_q=9.0), r1 + r2)
    c1, c2 = capi.Context(p, 0, lib_path), capi.Context(p, 0, lib_path)
    packs = [synth.pack(r1), synth.pack(r2)]
    for c, (seq, qual, off, ln) in zip((c1, c2), packs):
        c.submit_async(seq, qual, off[:-1].copy(), ln)
    seq, qual, off, ln = packs[0]
    pend = c1._pending
    try:
        c1.submit_async(seq, qual, off[:-1].copy(), ln)
        raise AssertionError("second pending batch was accepted")
    except capi.TgsfError as e:
        assert e.code == -1
    c1._pending = pend
    for c, (seq, qual, off, ln) in zip((c1, c2), packs):
        got_r, got_f = c.wait_result()
        exp_r, exp_f, exp_ctr = orc.filter_batch(p, seq, qual, off, ln, n_bins=c.n_bins)
        assert np.array_equal(got_r, exp_r) and np.array_equal(got_f, exp_f)
        assert np.array_equal(c.counters(), exp_ctr)
        c.close()


def align_windows_random(lib_path, n, seed=9, golden_dir=None, lengths=None, max_window=400, plant_whole=False):
    """tgsf_align_windows against edlib itself where oracle/_ref/libedlib_ref.so exists (compiled from the
    reference's include/edlib.cpp), else against the oracle's DP restatement: random adapters of 20..256 bp,
    windows of 5..400 bp with planted mutated copies, homopolymers and Ns, assorted k."""
    import ctypes as C
    ref_so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libedlib_ref.so")
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    adapters = [synth.ONT_RAPID, synth.PACBIO_BLUNT, b"AATGTACTTCGTTCAGTTACGTATTGCT", b"GCAATACGTAACTGAACGAAGT"]
    adapters += [bytes(acgt[rng.integers(0, 4, int(L))]) for L in (lengths or (20, 33, 64, 65, 90, 127, 128, 129, 150, 192, 193, 230, 256))]
    if lengths:                                   # (adapters beyond 256 bp: the wide path; a few short ones ride along)
        adapters = adapters[4:] + adapters[:2]
    p = abi.make_params("ont", adapters=adapters, max_batch_bases=1 << 22, max_batch_reads=4096 if not lengths else (256 if max_window <= 2600 else 16), max_read_len=4096,
                        **({"mid_match_len": 1, "end_match_len": 1} if lengths else {}))
    ctx = capi.Context(p, 0, lib_path)
    buf, off, ln, aid, ks, trip = bytearray(), [], [], [], [], []
    for i in range(n):
        a = int(rng.integers(0, len(adapters)))
        q = adapters[a]
        Q = len(q)
        T = int(rng.integers(5, 40)) if i % 9 == 0 else int(rng.integers(40, max_window))
        t = bytearray(acgt[rng.integers(0, 4, T)].tobytes())
        if i % 11 == 5:
            t = bytearray(b"T" * T)
        if plant_whole:                            # a whole (mutated) copy in a window that holds it: the path spans ~Q columns
            T = int(rng.integers(Q // 2, min(max_window, 2 * Q + 200)))
            t = bytearray(acgt[rng.integers(0, 4, T)].tobytes())
        for _ in range(int(rng.integers(0, 3)) if not plant_whole else 1):
            m = synth.mutate(rng, q, float(rng.choice([0.0, 0.05, 0.15, 0.3])))
            if plant_whole and rng.random() < 0.3:   # ... with a long stretch of other text in its middle
                h = len(m) // 2
                m = m[:h] + bytes(acgt[rng.integers(0, 4, int(rng.integers(1, 400)))]) + m[h:]
            if rng.random() < 0.3 and not plant_whole:
                m = m[int(rng.integers(0, len(m))):]
            pos = int(rng.integers(0, T))
            t[pos:pos + len(m)] = m
        if i % 13 == 7:
            t[int(rng.integers(0, len(t)))] = ord("N")
        t = bytes(t[:max_window])
        k = max(1, int(rng.choice([Q - 3, Q - 14, Q - 34, Q // 3, Q // 8 + 1])))
        k = min(k, Q - 1)
        off.append(len(buf)); ln.append(len(t)); aid.append(a); ks.append(k); trip.append((q, t, k))
        buf += t
    res, ends = ctx.align_windows(bytes(buf), off, ln, aid, ks)
    ctx.close()
    if os.path.exists(ref_so):
        class Cfg(C.Structure):
            _fields_ = [("k", C.c_int), ("mode", C.c_int), ("task", C.c_int), ("eq", C.c_void_p), ("neq", C.c_int)]

        class Res(C.Structure):
            _fields_ = [("status", C.c_int), ("editDistance", C.c_int), ("endLocations", C.POINTER(C.c_int)),
                        ("startLocations", C.POINTER(C.c_int)), ("numLocations", C.c_int),
                        ("alignment", C.POINTER(C.c_ubyte)), ("alignmentLength", C.c_int), ("alphabetLength", C.c_int)]
        lib = C.CDLL(ref_so)
        lib.edlibAlign.restype = Res
        lib.edlibAlign.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, Cfg]
        lib.edlibFreeAlignResult.argtypes = [Res]

        def truth(q, t, k):
            r = lib.edlibAlign(q, len(q), t, len(t), Cfg(k, 2, 2, None, 0))
            n_ = r.numLocations
            out = (r.editDistance, n_, r.alignmentLength, r.startLocations[0] if n_ else -1,
                   r.endLocations[0] if n_ else -1, r.endLocations[n_ - 1] if n_ else -1)
            lib.edlibFreeAlignResult(r)
            return out
    else:
        def truth(q, t, k):
            ed, n_, starts, ends_, alen = orc.align_hw(q, t, k)
            return (ed, n_, alen, starts[0] if n_ else -1, ends_[0] if n_ else -1, ends_[-1] if n_ else -1)
    bad = []
    for i, (q, t, k) in enumerate(trip):
        got = (int(res[i, 0]), int(res[i, 1]), int(res[i, 2]), int(res[i, 3]), int(ends[i, 0]), int(ends[i, 1]))
        exp = truth(q, t, k)
        if got != exp:
            bad.append((i, q, t, k, got, exp))
    assert not bad, "%d of %d alignments differ, first: %s" % (len(bad), n, bad[0])


def studded_reads(n_copies=90, gap=400, seed=21):
    """A read studded with exact adapter copies (more disjoint drop regions than the region kernel's in-register list
    holds) among ordinary ones: the reference completes such a run (src/TGSFilter.cpp:1376-1424 sort + merge)."""
    rng = np.random.default_rng(seed)
    reads = synth.make_reads(seed, 6, "ont", mean_len=3000, zoo=True, pmid=0.2)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for copies, ad in ((n_copies, synth.ONT_RAPID), (n_copies + 7, synth.ONT_RAPID_RC)):
        L = 600 + copies * gap
        seq = bytearray(acgt[rng.integers(0, 4, L)].tobytes())
        for c in range(copies):
            at = 400 + c * gap
            seq[at:at + len(ad)] = ad
        qual = bytes((np.clip(np.rint(rng.normal(18, 3, L)), 2, 40) + 33).astype(np.uint8))
        reads.append((b"studded%d" % copies, bytes(seq), qual))
    return reads


def many_regions(lib_path):
    reads = studded_reads()
    p = sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=8.0, min_len=100), reads)
    ctx = capi.Context(p, 0, lib_path)
    try:
        res, frags, ctr = compare_batch(ctx, p, reads)
        assert int(res["n_frags"].max()) > 64            # more kept fragments than the in-register region list holds
    finally:
        ctx.close()
