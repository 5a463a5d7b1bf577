#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs oracle/_ref/tgsfilter_ref, i.e. `make -C oracle ref`,
which compiles the reference from /root/reference where it lies).  The outputs are data:
  <case>.in.fq.gz      seeded synthetic input (tgsfilter_amd/synth.py); .in.bam / .in.sam.gz for the
                       BAM / SAM cases (written by tests/bamio.py from the same generator)
  <case>.out.fq.gz     the reference's clean FASTQ with -t 1 (byte-deterministic, input order)
  <case>.stderr.txt    the reference's stderr (INFO: lines = counters, resolved parameters)
  <case>.html.json     the <table> rows and the `var data = {...}` object of the HTML report
  <case>.cmd.json      the flags used
  edlib_vectors.json   kernel-level vectors from the reference's own edlib (HW/PATH)
No reference source text is stored.
"""
import ctypes as C
import gzip
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tgsfilter_amd import synth  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")
REF_EDLIB = os.path.join(ROOT, "oracle", "_ref", "libedlib_ref.so")

LONG_ADAPTER = (b"GGCGTCTGCTTGGGTGTTTAACCTTTTTGTCAGAGAGGTTCCAAGTCAGAGAGGTTCCT"
                b"ACGTTGCAATCGGATCCGATTACGGATCAAGT")  # 91 bp: two 64-row blocks

CASES = {
    # name: (synth kwargs, adapter fasta (list) or None, flags)
    "ont_zoo": (dict(seed=11, n=96, kind="ont", mean_len=4000, zoo=True, pmid=0.05),
                [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0"),
    "ont_trim": (dict(seed=12, n=96, kind="ont", mean_len=4000, zoo=True, pmid=0.05),
                 [synth.ONT_RAPID], "-x ont -l 500 -L 7000 -q 8 -Q 16 -5 12 -3 9 -T 30 -M 30 -m 6 -s 0.8 -S 0.85"),
    "ont_discard": (dict(seed=13, n=64, kind="ont", mean_len=4000, zoo=True, pmid=0.2),
                    [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -D"),
    "hifi_zoo": (dict(seed=14, n=96, kind="hifi", mean_len=3000, zoo=True, p5=0.2, p3=0.2, pmid=0.05),
                 [synth.PACBIO_BLUNT], "-x hifi -l 1000 -q 20 -5 0 -3 0"),
    "long_adapter": (dict(seed=15, n=64, kind="ont", mean_len=3000, zoo=True, adapter=LONG_ADAPTER,
                          pmid=0.1, err=0.05),
                     [LONG_ADAPTER, b"AATGTACTTCGTTCAGTTACGTATTGCT"], "-x ont -l 800 -q 9 -5 3 -3 0"),
    "ont_auto": (dict(seed=16, n=1500, kind="ont", mean_len=1300), None, "-x ont -l 1000 -b 8"),
    "hifi_auto": (dict(seed=17, n=1200, kind="hifi", mean_len=1400, p5=0.5, p3=0.4), None, "-x hifi -l 1000 -b 8"),
    "qc_only": (dict(seed=18, n=64, kind="ont", mean_len=3000, zoo=True), None, "--qc"),
    # repeat k-mer gate and length-ranked downsampling (config 5 flags)
    "ont_repeat": (dict(seed=19, n=80, kind="ont", mean_len=4500, zoo=True, pmid=0.05),
                   [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -p 3 -k 11"),
    "down_gd": (dict(seed=20, n=90, kind="ont", mean_len=3500, zoo=True, pmid=0.05),
                [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -g 40k -d 2"),
    "down_r": (dict(seed=21, n=90, kind="ont", mean_len=3500, zoo=True, pmid=0.05),
               [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -r 17 -p 2"),
    "down_R": (dict(seed=22, n=90, kind="hifi", mean_len=3000, p5=0.2, p3=0.2),
               [synth.PACBIO_BLUNT], "-x hifi -l 1000 -q 20 -5 0 -3 0 -R 0.3"),
    "down_F": (dict(seed=23, n=70, kind="ont", mean_len=3000), None, "-F -r 20"),
    # unaligned BAM / SAM input (the reference reads them through htslib); a few bases are
    # IUPAC / odd characters, which its 4-bit round trip turns into NUL or N
    "hifi_bam": (dict(seed=24, n=60, kind="hifi", mean_len=3000, zoo=True, p5=0.2, p3=0.2, pmid=0.05),
                 [synth.PACBIO_BLUNT], "-x hifi -l 1000 -q 20 -5 0 -3 0", "bam"),
    "ont_sam": (dict(seed=25, n=60, kind="ont", mean_len=3000, zoo=True, pmid=0.05),
                [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0", "sam"),
    "hifi_bam_auto": (dict(seed=26, n=700, kind="hifi", mean_len=1400, p5=0.5, p3=0.4), None, "-x hifi -l 1000 -b 8", "bam"),
    # FASTA input (records without qualities): count-only tallies, soft-masked stretches at the read ends
    "ont_fasta": (dict(seed=27, n=80, kind="ont", mean_len=3500, zoo=True, pmid=0.05),
                  [synth.ONT_RAPID], "-x ont -l 1000 -5 0 -3 0", "fa"),
    "hifi_fasta_auto": (dict(seed=28, n=700, kind="hifi", mean_len=1400, p5=0.5, p3=0.4), None, "-x hifi -l 1000 -b 8", "fa"),
    "fasta_down": (dict(seed=29, n=70, kind="ont", mean_len=3000, zoo=True, pmid=0.05),
                   [synth.ONT_RAPID], "-x ont -l 1000 -5 4 -3 0 -r 15 -p 2", "fa"),
}

HUGE_A = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(1501).integers(0, 4, 150)])   # 3 blocks
HUGE_B = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(2301).integers(0, 4, 230)])   # 4 blocks
CASES.update({
    # parameter domain beyond the defaults: edlib's k >= Q corner (-m 1 / -M 1; read ends made of
    # N only, where no adapter character matches and edlib reports the location -1), adapters of 3 and 4 blocks
    "ont_m1": (dict(seed=30, n=96, kind="ont", mean_len=3500, zoo=True, pmid=0.1),
               [synth.ONT_RAPID], "-x ont -l 500 -q 9 -5 0 -3 0 -m 1 -M 1", "fq", "n_ends"),
    "huge_adapter": (dict(seed=31, n=72, kind="ont", mean_len=3500, zoo=True, adapter=HUGE_A, pmid=0.15, err=0.06),
                     [HUGE_A, HUGE_B], "-x ont -l 800 -q 9 -5 0 -3 2"),
    # the repeat gate beyond k = 13 (64-bit k-mers; at k = 32 the reference's mask (1ULL << 64) - 1 is what the
    # machine makes of it, :1748)
    "repeat_k15": (dict(seed=34, n=70, kind="ont", mean_len=4000, zoo=True, pmid=0.05), [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -p 3 -k 15"),
    "repeat_k21": (dict(seed=35, n=70, kind="ont", mean_len=4000, zoo=True, pmid=0.05), [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -p 2 -k 21"),
    "repeat_k32": (dict(seed=36, n=70, kind="ont", mean_len=4000, zoo=True, pmid=0.05), [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -p 2 -k 32"),
    "repeat_k32b": (dict(seed=36, n=70, kind="ont", mean_len=4000, zoo=True, pmid=0.05), [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -p 3000 -k 32"),
    # Phred+64 qualities through the whole command line (Get_qType :1042-1077 decides from the first reads)
    "ont_phred64": (dict(seed=32, n=90, kind="ont", mean_len=3000, zoo=True, pmid=0.05),
                    [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0", "fq", "phred64"),
    "hifi_phred64_auto": (dict(seed=33, n=900, kind="hifi", mean_len=1400, p5=0.5, p3=0.4), None, "-x hifi -l 1000 -b 8", "fq", "phred64"),
})

WIDE_A = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(3001).integers(0, 4, 300)])
WIDE_B = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(7001).integers(0, 4, 700)])
CASES.update({
    # -e beyond the 512 positions one launch of the end-table kernel tallies (round 3: one launch per slab of 512)
    "ont_e1300": (dict(seed=37, n=80, kind="ont", mean_len=3500, zoo=True, pmid=0.05), [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 0 -3 0 -e 1300"),
    # adapters beyond 256 bp (-a accepts any length; the reference's edlib is multi-block): 300 and 700 bp, five and eleven words
    "wide_adapter": (dict(seed=38, n=60, kind="ont", mean_len=4500, zoo=True, adapter=WIDE_A, pmid=0.3, err=0.05),
                     [WIDE_A, WIDE_B], "-x ont -l 800 -q 9 -5 0 -3 2 -M 120"),
})

CASES.update({
    # quality bytes of 128 and above: the reference subtracts qType from a (signed) char (:1455-1457, :1508) -- such a byte
    # stands for its value - 256; every read keeps a mean in [0, 256) (outside it the reference indexes out of bounds, :1943)
    "ont_high_qual": (dict(seed=39, n=90, kind="ont", mean_len=3000, zoo=True, pmid=0.05), [synth.ONT_RAPID], "-x ont -l 1000 -q 10 -5 6 -3 3", "fq", "high_bytes"),
})

GIANT_A = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(20481).integers(0, 4, 2048)])
GIANT_B = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(50001).integers(0, 4, 5000)])
CASES.update({
    # adapters beyond 1 280 bp (round 5): where the traceback state of the first location reaches 1 MiB (from ~1 800 bp) edlib
    # finds its path by Hirschberg's divide and conquer (include/edlib.cpp:1191-1210, 1234-1400): 2 048 and 5 000 bp, planted
    # at the ends and in the middle of reads of 12-16 kb with 0-10 % errors and long insertions
    "giant_adapter": (dict(seed=40, n=9, kind="ont", mean_len=16000, p5=0.0, pmid=0.0), [GIANT_A, GIANT_B],
                      "-x ont -l 1000 -q 7 -5 0 -3 2 -M 1200 -m 900 -E 2600 -T 40", "fq", "giant"),
})

IN_EXT = {"fq": "in.fq", "bam": "in.bam", "sam": "in.sam", "fa": "in.fa"}


def tweak(reads, how):
    out = []
    if how == "giant":
        rng = np.random.default_rng(4040)
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        for i, (name, sq, q) in enumerate(reads):
            L = int(rng.integers(12000, 16000))
            sq = bytearray(acgt[rng.integers(0, 4, L)].tobytes())
            q = bytes((rng.integers(12, 35, L) + 33).astype(np.uint8))
            ad = GIANT_A if i % 2 == 0 else GIANT_B
            m = synth.mutate(rng, ad, float(rng.choice([0.0, 0.03, 0.1])))
            if i % 4 == 3:
                h = len(m) // 2
                m = m[:h] + bytes(acgt[rng.integers(0, 4, int(rng.integers(1, 300)))]) + m[h:]
            pos = int(rng.integers(0, 40)) if i % 3 == 0 else (int(rng.integers(2700, L - 2700 - len(m))) if i % 3 == 1 else L - len(m) - int(rng.integers(0, 40)))
            if i % 9 != 8:
                sq[pos:pos + len(m)] = m
            out.append((name, bytes(sq), q))
        return out
    for i, (name, sq, q) in enumerate(reads):
        if how == "phred64":
            q = bytes(min(max(c, 33 + 16) + 31, 126) for c in q)     # every quality >= 16: min char 80 > 78 decides Phred64 (:1050)
        elif how == "high_bytes" and i % 3 != 2 and len(sq) > 400:
            rng = np.random.default_rng(1000 + i)
            qa = bytearray(max(c, 33 + 30) for c in q)
            if i % 3 == 0:
                for pos in list(rng.integers(0, len(qa), max(1, len(qa) // 300))) + [0, len(qa) - 1, 99, 100]:
                    qa[int(pos)] = int(rng.integers(128, 256))
            else:
                a = int(rng.integers(0, len(qa) - 210))
                for j in range(a, a + 205):
                    qa[j] = 255 if j % 3 else 128
                qa = bytearray(c if c >= 128 else 126 for c in qa)
            q = bytes(qa)
        elif how == "n_ends" and i % 5 == 2 and len(sq) > 900:
            sq = b"N" * 260 + sq[260:-260] + b"N" * 260
        out.append((name, sq, q))
    return out



def soft_mask(reads, seed):
    """Lower-case stretches over both ends and the middle of every other read (FASTA cases)."""
    rng = np.random.default_rng(seed)
    out = []
    for i, (name, s, q) in enumerate(reads):
        b = bytearray(s)
        if i % 2 == 0:
            for lo, hi in ((0, 60), (len(b) - 60, len(b)), (len(b) // 2, len(b) // 2 + 80)):
                for k in range(max(lo, 0), min(hi, len(b))):
                    if rng.random() < 0.7 and chr(b[k]).isalpha():
                        b[k] |= 0x20
        out.append((name, bytes(b), q))
    return out


def odd_bases(reads, seed):
    """Sprinkle IUPAC codes, '=', digits and other characters over a few reads (SAM/BAM cases)."""
    rng = np.random.default_rng(seed)
    odd = np.frombuffer(b"RYKMSWBDHVU=.x0123nacgt", dtype=np.uint8)
    out = []
    for i, (name, s, q) in enumerate(reads):
        if i % 4 == 1:
            b = bytearray(s)
            for pos in rng.integers(0, len(b), max(1, len(b) // 300)):
                b[int(pos)] = int(odd[rng.integers(0, len(odd))])
            s = bytes(b)
        out.append((name, s, q))
    return out


def html_slices(html: str):
    table = re.findall(r"<tr>.*?</tr>", html, flags=re.S)
    # the data object only: it ends at the "}" right before </script> (nothing of the page's
    # own script text is kept)
    m = re.search(r"var data = (\{.*?\})\n</script>", html, flags=re.S)
    # the whole document with the time stamp blanked (include/report.cpp:108), as a digest
    import hashlib
    doc = re.sub(r"\d{4}-\d\d-\d\d \d\d:\d\d:\d\d", "T", html)
    return {"table_rows": table, "data": m.group(1) if m else None, "doc_sha256": hashlib.sha256(doc.encode("utf-8")).hexdigest()}


def run_case(name, kwargs, adapters, flags, fmt="fq", how=None):
    reads = synth.make_reads(**kwargs)
    if how:
        reads = tweak(reads, how)
    with tempfile.TemporaryDirectory() as td:
        fin = os.path.join(td, IN_EXT[fmt])
        fout = os.path.join(td, "out.fa" if fmt == "fa" else "out.fq")
        if fmt == "fq":
            synth.write_fastq(fin, reads)
        elif fmt == "fa":
            with open(fin, "wb") as f:
                for rname, sq, _ in soft_mask(reads, kwargs["seed"]):
                    f.write(b">" + rname + b"\n" + sq + b"\n")
        else:
            from tests import bamio
            reads = odd_bases(reads, kwargs["seed"])
            if fmt == "bam":
                bamio.write_bam(fin, reads, block=0x4000)
            else:
                bamio.write_sam(fin, reads)
        cmd = [REF_BIN, "-i", fin, "-t", "1"] + flags.split()
        if "--qc" not in flags:
            cmd += ["-o", fout]
        if adapters:
            fa = os.path.join(td, "adapters.fa")
            with open(fa, "wb") as f:
                for i, a in enumerate(adapters):
                    f.write(b">a%d\n" % i + a + b"\n")
            cmd += ["-a", fa]
        p = subprocess.run(cmd, capture_output=True, cwd=td)
        stderr = p.stderr.decode().replace(td + "/", "")
        out = open(fout, "rb").read() if os.path.exists(fout) else b""
        htmlname = os.path.join(td, "out.html" if "--qc" not in flags else "in.html")
        html = open(htmlname, encoding="utf-8", errors="replace").read() if os.path.exists(htmlname) else ""
        raw = open(fin, "rb").read()
    if fmt == "bam":
        open(os.path.join(HERE, name + ".in.bam"), "wb").write(raw)        # already BGZF-compressed
    else:
        with gzip.GzipFile(os.path.join(HERE, name + "." + IN_EXT[fmt] + ".gz"), "wb", mtime=0) as f:
            f.write(raw)
    with gzip.GzipFile(os.path.join(HERE, name + ".out.fq.gz"), "wb", mtime=0) as f:
        f.write(out)
    open(os.path.join(HERE, name + ".stderr.txt"), "w").write(stderr)
    json.dump(html_slices(html), open(os.path.join(HERE, name + ".html.json"), "w"), indent=0)
    json.dump({"flags": flags, "adapters": [a.decode() for a in adapters] if adapters else None,
               "synth": kwargs if "adapter" not in kwargs else {**kwargs, "adapter": kwargs["adapter"].decode()},
               "returncode": p.returncode, "in_format": fmt},
              open(os.path.join(HERE, name + ".cmd.json"), "w"), indent=1)
    print(name, "rc", p.returncode, "in", len(raw), "out", len(out))
    print("   ", "\n    ".join(l for l in stderr.splitlines() if "reads" in l or "adapter" in l or "trim" in l))


class _Cfg(C.Structure):
    _fields_ = [("k", C.c_int), ("mode", C.c_int), ("task", C.c_int), ("eq", C.c_void_p), ("neq", C.c_int)]


class _Res(C.Structure):
    _fields_ = [("status", C.c_int), ("editDistance", C.c_int), ("endLocations", C.POINTER(C.c_int)),
                ("startLocations", C.POINTER(C.c_int)), ("numLocations", C.c_int),
                ("alignment", C.POINTER(C.c_ubyte)), ("alignmentLength", C.c_int), ("alphabetLength", C.c_int)]


def edlib_vectors(n=600, seed=7):
    """Kernel-level vectors: the reference's edlibAlign(HW=2, PATH=2) on seeded triples."""
    lib = C.CDLL(REF_EDLIB)
    lib.edlibAlign.restype = _Res
    lib.edlibAlign.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, _Cfg]
    lib.edlibFreeAlignResult.argtypes = [_Res]
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    lib22 = [synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC, synth.ONT_RAPID, synth.ONT_RAPID_RC,
             b"GCAATACGTAACTGAACGAAGT", LONG_ADAPTER,
             b"GGAACCTCTCTGACTTGGAACCTCTCTGACAAAAAGGTTAAACACCCAAGCAGACGCCAGCAAT"]
    vec = []
    for i in range(n):
        q = lib22[int(rng.integers(0, len(lib22)))]
        Q = len(q)
        T = int(rng.integers(5, 30)) if i % 7 == 0 else int(rng.integers(30, 400))
        t = bytearray(acgt[rng.integers(0, 4, T)].tobytes())
        if i % 5 == 4:
            t = bytearray(b"A" * T)
        for _ in range(int(rng.integers(0, 3))):
            m = synth.mutate(rng, q, float(rng.choice([0.0, 0.05, 0.1, 0.2, 0.3])))
            if rng.random() < 0.3:
                m = m[int(rng.integers(0, len(m))):]
            p = int(rng.integers(0, T))
            t[p:p + len(m)] = m
        t = bytes(t[:max(5, min(len(t), 400))])
        k = int(rng.choice([Q - 3, Q - 34, Q - 14, Q // 3]))
        k = max(0, k)
        r = lib.edlibAlign(q, Q, t, len(t), _Cfg(k, 2, 2, None, 0))
        vec.append({"q": q.decode(), "t": t.decode(), "k": k, "ed": r.editDistance, "n": r.numLocations,
                    "starts": [r.startLocations[j] for j in range(r.numLocations)],
                    "ends": [r.endLocations[j] for j in range(r.numLocations)],
                    "alen": r.alignmentLength})
        lib.edlibFreeAlignResult(r)
    json.dump(vec, open(os.path.join(HERE, "edlib_vectors.json"), "w"))
    print("edlib vectors", len(vec), "hits", sum(v["ed"] >= 0 for v in vec))


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, case in CASES.items():
        if not only or name in only:
            run_case(name, *case)
    if not only or "edlib" in only:
        edlib_vectors()
