"""Seeded synthetic long-read generator (SURVEY.md §8d shapes).

Used by the tests, by the golden-vector script and by bench.py.  Everything is
derived from ``numpy.random.Generator(PCG64(seed))`` so that every box with this
image regenerates identical bytes.  Nothing here touches the GPU or the oracle.

The adapter library is the reference's table of 22 sequences
(src/TGSFilter.cpp:2970-2991); only the entries the configs need are named.
"""
from __future__ import annotations

import numpy as np

# src/TGSFilter.cpp:2970-2971, :2978-2979
PACBIO_BLUNT = b"ATCTCTCTCTTTTCCTCCTCCTCCGTTGTTGTTGTTGAGAGAGAT"
PACBIO_BLUNT_RC = b"ATCTCTCTCAACAACAACAACGGAGGAGGAGGAAAAGAGAGAGAT"
ONT_RAPID = b"GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"
ONT_RAPID_RC = b"TGAAGCGGCGCACGAAAAACGCGAAAGCGTTTCACGATAAATGCGAAAAC"

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.full(256, ord("N"), dtype=np.uint8)
for _a, _b in zip(b"ACGTacgtMRWSYKmrwsyk", b"TGCAtgcaKYWSRMkywsrm"):
    _COMP[_a] = _b


def revcomp(s: bytes) -> bytes:
    """rev_comp_seq, src/TGSFilter.cpp:859-867 with the table of :2954-2967."""
    return _COMP[np.frombuffer(s, dtype=np.uint8)][::-1].tobytes()


def mutate(rng: np.random.Generator, s: bytes, rate: float) -> bytes:
    """Per-base substitution / insertion / deletion at total rate ``rate``."""
    out = bytearray()
    for ch in s:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(int(_ACGT[rng.integers(0, 4)]))
            out.append(ch)
            continue
        if u < rate:
            out.append(int(_ACGT[rng.integers(0, 4)]))
            continue
        out.append(ch)
    return bytes(out)


def _qual(rng, n, mean, sd, lo, hi, offset=33):
    q = np.clip(np.rint(rng.normal(mean, sd, n)), lo, hi).astype(np.uint8)
    return (q + offset).astype(np.uint8)


def ont_lengths(rng, n, mean=45000.0, sigma=0.8, lo=200, hi=2_000_000):
    mu = np.log(mean) - 0.5 * sigma * sigma
    return np.clip(rng.lognormal(mu, sigma, n), lo, hi).astype(np.int64)


def hifi_lengths(rng, n, mean=18000.0, sd=3000.0, lo=1000, hi=40000):
    return np.clip(rng.normal(mean, sd, n), lo, hi).astype(np.int64)


def make_reads(seed: int, n: int, kind: str = "ont", *, mean_len: float | None = None,
               max_len: int | None = None, p5: float | None = None, p3: float | None = None,
               pmid: float | None = None, adapter: bytes | None = None, err: float | None = None,
               zoo: bool = False):
    """Return a list of (name, seq, qual) byte triples.

    kind="ont": config C2 shape -- lognormal lengths, per-read mean Q from
    {7,9,12,14,18}, rapid adapter at the 5' end of 80 % of reads (0-30 random
    bases before it, 10 % errors), 0.03 % middle.
    kind="hifi": config C1/C3 shape -- N(18k,3k) lengths, Q~N(30,6), blunt
    adapter 5' 0.27 %, 3' 0.26 %, middle 0.002 %, 3 % errors.
    ``zoo=True`` additionally plants the edge cases SURVEY §8c lists.
    """
    rng = np.random.default_rng(seed)
    if kind == "ont":
        lens = ont_lengths(rng, n, mean_len or 45000.0)
        adapter = adapter or ONT_RAPID
        p5 = 0.80 if p5 is None else p5
        p3 = 0.0 if p3 is None else p3
        pmid = 0.0003 if pmid is None else pmid
        err = 0.10 if err is None else err
    elif kind == "hifi":
        lens = hifi_lengths(rng, n, mean_len or 18000.0,
                            sd=(mean_len or 18000.0) / 6.0,
                            lo=min(1000, int((mean_len or 18000) // 2)))
        adapter = adapter or PACBIO_BLUNT
        p5 = 0.0027 if p5 is None else p5
        p3 = 0.0026 if p3 is None else p3
        pmid = 0.00002 if pmid is None else pmid
        err = 0.03 if err is None else err
    else:
        raise ValueError(kind)
    if max_len is not None:
        lens = np.minimum(lens, max_len)
    reads = []
    for i in range(n):
        L = int(lens[i])
        seq = bytearray(_ACGT[rng.integers(0, 4, L)].tobytes())
        if kind == "ont":
            mq = float(rng.choice([7, 9, 12, 14, 18]))
            qual = bytearray(_qual(rng, L, mq, 4.0, 1, 50).tobytes())
        else:
            qual = bytearray(_qual(rng, L, 30.0, 6.0, 2, 60).tobytes())
        u5, u3, um = rng.random(), rng.random(), rng.random()
        if u5 < p5:
            a = mutate(rng, adapter, err)
            pre = int(rng.integers(0, 31))
            if pre + len(a) < L:
                seq[pre:pre + len(a)] = a
        if u3 < p3:
            a = mutate(rng, revcomp(adapter), err)
            pre = int(rng.integers(0, 31))
            if pre + len(a) < L:
                seq[L - pre - len(a):L - pre] = a
        if um < pmid and L > 1000:
            a = mutate(rng, adapter if rng.random() < 0.5 else revcomp(adapter), err / 2)
            p = int(rng.integers(300, L - 300 - len(a)))
            seq[p:p + len(a)] = a
        if zoo:
            _zoo(rng, i, seq, qual, adapter, L)
        seq = bytes(seq[:len(qual)]) if len(seq) > len(qual) else bytes(seq)
        q = bytes(qual[:len(seq)])
        name = b"read%d len=%d" % (i, len(seq)) if i % 3 else b"read%d" % i
        reads.append((name, seq, q))
    return reads


def _zoo(rng, i, seq, qual, adapter, L):
    """Edge cases of SURVEY §8c, planted on a rotating schedule."""
    k = i % 16
    rc = revcomp(adapter)
    if k == 1 and L > 3000:            # two middle adapters -> split into 3 fragments
        for frac in (0.33, 0.66):
            p = int(L * frac)
            seq[p:p + len(adapter)] = adapter
    elif k == 2 and L > 1500:          # middle adapter close to the end -> short tail fragment
        p = L - 700
        seq[p:p + len(rc)] = rc
    elif k == 3 and L > 4000:          # fragment failing Q after split
        p = L // 2
        seq[p:p + len(adapter)] = mutate(rng, adapter, 0.02)[:len(adapter)]
        qual[p + 60:] = bytes([33 + 3]) * (len(qual) - p - 60)
    elif k == 4:                        # N and lower-case bases
        for p in rng.integers(0, L, 40):
            seq[int(p)] = ord("N")
        for p in rng.integers(0, L, 40):
            seq[int(p)] = seq[int(p)] | 0x20
    elif k == 5:                        # homopolymer-rich ends: many equal-best locations
        n = min(L // 3, 120)
        seq[:n] = b"A" * n
        seq[L - n:] = b"T" * n
    elif k == 6:                        # exact adapter copies at both ends
        if L > 200:
            seq[:len(adapter)] = adapter
            seq[L - len(rc):] = rc
    elif k == 7:                        # very short reads: windows clamp, no middle search
        n = int(rng.integers(5, 260))
        del seq[n:]
        del qual[n:]
    elif k == 8 and L > 2000:           # adapter straddling the end/middle boundary (E=150)
        seq[130:130 + len(adapter)] = adapter
        seq[L - 130 - len(rc):L - 130] = rc
    elif k == 9 and L > 2000:           # tandem adapter copies in the middle
        p = L // 2
        seq[p:p + 2 * len(adapter)] = adapter + adapter
    elif k == 10:                       # low quality read
        qual[:] = bytes([33 + 4]) * len(qual)
    elif k == 11 and L > 1200:          # partial adapter (prefix only) at the 5' end
        seq[:20] = adapter[-20:]
    elif k == 12 and L > 5000:          # heavily mutated middle adapter (borderline similarity)
        p = L // 3
        a = mutate(rng, adapter, 0.12)
        seq[p:p + len(a)] = a


def write_fastq(path, reads):
    with open(path, "wb") as f:
        for name, seq, qual in reads:
            f.write(b"@" + name + b"\n" + seq + b"\n+\n" + qual + b"\n")


def read_fastq(path):
    """Strict 4-line FASTQ as FastxReader::readFastq keeps it (src/TGSFilter.cpp:685-726)."""
    import gzip
    op = gzip.open if str(path).endswith(".gz") else open
    out = []
    with op(path, "rb") as f:
        data = f.read().split(b"\n")
    for i in range(0, len(data) - 3, 4):
        out.append((data[i][1:], data[i + 1], data[i + 3]))
    return out


def pack(reads, align: int = 16):
    """CSR-pack (name, seq, qual) triples: returns seq, qual (uint8), offsets (uint64, n+1,
    each read start aligned to ``align`` bytes), lengths (uint32)."""
    n = len(reads)
    lengths = np.fromiter((len(r[1]) for r in reads), dtype=np.uint32, count=n)
    padded = (lengths.astype(np.uint64) + np.uint64(align - 1)) // np.uint64(align) * np.uint64(align)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(padded, out=offsets[1:])
    total = int(offsets[-1])
    seq = np.zeros(total, dtype=np.uint8)
    qual = np.zeros(total, dtype=np.uint8)
    for i, (_, s, q) in enumerate(reads):
        o = int(offsets[i])
        seq[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
        qual[o:o + len(q)] = np.frombuffer(q, dtype=np.uint8)
    return seq, qual, offsets, lengths


# ---------------------------------------------------------------------------
# C2-shaped FASTQ text, written by several processes (bench.py's end-to-end leg: GBs of text in seconds)
# ---------------------------------------------------------------------------
def _fastq_chunk(job):
    """One run of reads -> its byte range of the file.  Same distributions as make_reads(kind='ont')."""
    path, file_off, first, lens, seed = job[:5]
    hifi = len(job) > 5 and job[5] == "hifi"
    rng = np.random.default_rng(seed)
    n = len(lens)
    names = [b"@r%d\n" % (first + i) for i in range(n)]
    sizes = np.array([len(nm) for nm in names], dtype=np.int64) + 2 * lens + 4
    starts = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(sizes, out=starts[1:])
    total = int(starts[-1])
    buf = _ACGT[rng.integers(0, 4, total, dtype=np.uint8)]
    B = int(lens.sum())
    mq = rng.choice(np.array([30], dtype=np.float32) if hifi else np.array([7, 9, 12, 14, 18], dtype=np.float32), n)
    q = rng.standard_normal(B, dtype=np.float32)
    q *= 6.0 if hifi else 4.0
    q += np.repeat(mq, lens)
    np.rint(q, out=q)
    np.clip(q, 2 if hifi else 1, 60 if hifi else 50, out=q)
    q += 33
    q8 = q.astype(np.uint8)
    del q
    at = 0
    for i in range(n):
        L = int(lens[i])
        s = int(starts[i])
        nm = names[i]
        buf[s:s + len(nm)] = np.frombuffer(nm, dtype=np.uint8)
        s0 = s + len(nm)
        if hifi:                                      # blunt adapter: 5' 0.27 %, 3' 0.26 %, middle 0.002 % of reads, 3 % errors
            u = rng.random()
            if u < 0.0027 + 0.0026 + 0.00002 and L > 2000:
                a = np.frombuffer(mutate(rng, PACBIO_BLUNT, 0.03), dtype=np.uint8)
                p = 0 if u < 0.0027 else (L - len(a) if u < 0.0053 else int(rng.integers(300, L - 300 - len(a))))
                buf[s0 + p:s0 + p + len(a)] = a
        elif rng.random() < 0.80:                     # rapid adapter at the 5' end, 0-30 bases in, 10 % errors
            a = np.frombuffer(mutate(rng, ONT_RAPID, 0.10), dtype=np.uint8)
            pre = int(rng.integers(0, 31))
            if pre + len(a) < L:
                buf[s0 + pre:s0 + pre + len(a)] = a
        if not hifi and rng.random() < 0.0003 and L > 2000:        # 0.03 % in the middle
            a = np.frombuffer(mutate(rng, ONT_RAPID if rng.random() < 0.5 else ONT_RAPID_RC, 0.05), dtype=np.uint8)
            p = int(rng.integers(300, L - 300 - len(a)))
            buf[s0 + p:s0 + p + len(a)] = a
        buf[s0 + L:s0 + L + 3] = (10, 43, 10)         # "\n+\n"
        buf[s0 + L + 3:s0 + 2 * L + 3] = q8[at:at + L]
        buf[s0 + 2 * L + 3] = 10
        at += L
    with open(path, "r+b") as f:
        f.seek(file_off)
        f.write(memoryview(buf))
    return B


def write_ont_fastq(path, n_reads, seed=2, mean_len=45000.0, max_len=2_000_000, procs=None, reads_per_job=256, kind="ont"):
    """FASTQ text of config C2's shape (kind="hifi": C1/C3's, mean_len 18000) at `path`; returns
    (bases, file bytes).  Deterministic in (n_reads, seed, mean_len, max_len, reads_per_job), whatever the number of
    processes."""
    import multiprocessing as mp
    import os
    rng = np.random.default_rng(seed)
    if kind == "hifi":
        lens = np.minimum(hifi_lengths(rng, n_reads, mean_len, sd=mean_len / 6.0), max_len).astype(np.int64)
    else:
        lens = np.minimum(ont_lengths(rng, n_reads, mean_len), max_len).astype(np.int64)
    name_len = np.array([len(b"@r%d\n" % i) for i in range(n_reads)], dtype=np.int64)
    sizes = name_len + 2 * lens + 4
    jobs, off = [], 0
    for a in range(0, n_reads, reads_per_job):
        b = min(n_reads, a + reads_per_job)
        jobs.append((path, off, a, lens[a:b], seed * 1_000_003 + a, kind))
        off += int(sizes[a:b].sum())
    with open(path, "wb") as f:
        f.truncate(off)
    procs = procs or max(1, min(32, (os.cpu_count() or 2) - 1))
    if procs == 1:
        done = [_fastq_chunk(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            done = pool.map(_fastq_chunk, jobs, chunksize=1)
    return int(sum(done)), off
