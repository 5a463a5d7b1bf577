"""The command line against the REFERENCE BINARY run side by side on freshly generated inputs and random flag
sets (oracle/_ref/tgsfilter_ref: compiled from the reference where it lies, by `make -C oracle ref`; it travels
to the GPU box with the repo).  Complements the committed goldens: nothing here was seen when the code was written.
CPU: through the emulation build of the CLI (small inputs).  GPU (-m gpu): the real binary."""
import os

import numpy as np
import pytest

from tests import cli_check
from tgsfilter_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")


def case(seed, n):
    rng = np.random.default_rng(seed)
    kind = "ont" if rng.random() < 0.6 else "hifi"
    fasta = rng.random() < 0.15
    scale = float(os.environ.get("TGSF_LIVE_LEN_SCALE", "1"))          # campaigns with long reads (tests/manual/live_campaign.py)
    reads = synth.make_reads(int(rng.integers(1, 1 << 30)), n, kind, mean_len=scale * float(rng.choice([1500, 3000, 5000])),
                             zoo=bool(rng.random() < 0.7), pmid=float(rng.choice([0.0, 0.05, 0.2])))
    flags = ["-x", kind, "-l", str(int(rng.choice([500, 1000, 2000])))]
    if rng.random() < 0.5 and not fasta:
        flags += ["-q", str(int(rng.choice([7, 10, 13]) if kind == "ont" else rng.choice([15, 20, 28])))]
    if rng.random() < 0.3 and not fasta:
        flags += ["-Q", str(int(rng.choice([16, 30]) if kind == "ont" else rng.choice([33, 45])))]
    if rng.random() < 0.3:
        flags += ["-L", str(int(rng.choice([4000, 9000])))]
    if rng.random() < 0.6:
        flags += ["-5", str(int(rng.choice([0, 0, 6, 25]))), "-3", str(int(rng.choice([0, 0, 9])))]
    else:
        flags += ["-b", str(int(rng.choice([2, 8]))), "-n", "500"]
    # With automatic trims and a check length above -e the reference is not deterministic (its 3' base-content
    # thread clamps whatever the 5' thread has stored by then, src/TGSFilter.cpp:1135-1137): keep -E <= -e there.
    auto_trims = "-b" in flags
    if rng.random() < 0.3:
        flags += ["-E", str(int(rng.choice([100, 140]) if auto_trims else rng.choice([100, 250])))]
    if rng.random() < 0.3:
        flags += ["-e", str(int(rng.choice([150, 220]) if auto_trims else rng.choice([80, 150, 220])))]
    if rng.random() < 0.3:
        flags += ["-m", str(int(rng.choice([6, 12, 1]))), "-M", str(int(rng.choice([25, 30, 1])))]     # 1: edlib's k >= Q corner
    if rng.random() < 0.3:
        flags += ["-T", str(int(rng.choice([0, 20])))]
    if rng.random() < 0.3:
        flags += ["-s", str(float(rng.choice([0.8, 0.85]))), "-S", str(float(rng.choice([0.85, 0.92])))]
    if rng.random() < 0.2:
        flags += ["-D"]
    if rng.random() < float(os.environ.get("TGSF_LIVE_GATE_P", "0.2")):       # (a campaign can ask for the repeat gate more often)
        flags += ["-p", str(int(rng.choice([2, 5]))), "-k", str(int(rng.choice([9, 11, 12, 13, 15, 16, 24, 31, 32])))]
    if rng.random() < 0.2:
        flags += ["-r", str(int(rng.integers(3, n)))]
    adapters = None
    if rng.random() < 0.7:
        adapters = [synth.ONT_RAPID if kind == "ont" else synth.PACBIO_BLUNT]
        if rng.random() < 0.35:                      # several adapters (passes of up to 4), one of them two words long
            extra = [b"AATGTACTTCGTTCAGTTACGTATTGCT", b"GCAATACGTAACTGAACGAAGT",
                     b"GGCGTCTGCTTGGGTGTTTAACCTTTTTGTCAGAGAGGTTCCAAGTCAGAGAGGTTCCTACGTTGCAATCGGATCCGATTACGGATCAAGT",
                     b"CTTGCGGGCGGCGGACTCTCCTCTGAAGATAGAGCGACAGGCAAG",
                     bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(77).integers(0, 4, 170)])]   # three words
            for i in rng.choice(len(extra), int(rng.integers(1, 3)), replace=False):
                adapters.append(extra[int(i)])
    return reads, flags, adapters, fasta


def case2(seed, n):
    """Wider surface: input formats (gz, SAM, BAM, FASTA), output forms (-f, .gz), --qc, -F, -A-less downsampling."""
    rng = np.random.default_rng(seed)
    reads, flags, adapters, fasta = case(seed + 7919, n)
    in_fmt = "fa" if fasta else str(rng.choice(["fq", "fq", "fq.gz", "bam", "sam"]))
    if in_fmt in ("bam", "sam"):
        reads = [(name.split()[0], s, q) for name, s, q in reads]
    out_name = "out.fa" if in_fmt == "fa" else "out.fq"
    u = rng.random()
    if u < 0.1:
        flags = ["--qc"]
        adapters = None
    elif u < 0.2:
        flags = ["-F", "-r", str(int(rng.integers(3, n)))]
        adapters = None
    elif u < 0.3 and "-r" not in flags:
        flags += ["-R", str(float(rng.choice([0.2, 0.5])))]
    elif u < 0.4 and "-r" not in flags:
        flags += ["-g", str(int(rng.choice([20, 60]))) + "k", "-d", str(int(rng.choice([2, 5])))]
    v = rng.random()
    if v < 0.15 and in_fmt != "fa":
        flags += ["-f"]
        out_name = "out.fa"
    elif v < 0.3:
        out_name += ".gz"
        if rng.random() < 0.5:
            flags += ["-c", str(int(rng.choice([1, 6, 9])))]
    return reads, flags, adapters, in_fmt, out_name


needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")


@needs_ref
@pytest.mark.parametrize("seed", range(300, 312))
def test_cli_live_emul(seed):
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), "emul"], check=True)
    reads, flags, adapters, fasta = case(seed, 60)
    cli_check.compare_live(os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"), REF, reads, flags, adapters, fasta)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(300, 324))
def test_cli_live_gpu(seed):
    reads, flags, adapters, fasta = case(seed, 400)
    cli_check.compare_live(os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter"), REF, reads, flags, adapters, fasta)


@needs_ref
@pytest.mark.parametrize("seed", range(700, 716))
def test_cli_live_formats_emul(seed):
    reads, flags, adapters, in_fmt, out_name = case2(seed, 60)
    cli_check.compare_live(os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"), REF, reads, flags, adapters,
                           in_fmt=in_fmt, out_name=out_name)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(700, 724))
def test_cli_live_formats_gpu(seed):
    reads, flags, adapters, in_fmt, out_name = case2(seed, 300)
    cli_check.compare_live(os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter"), REF, reads, flags, adapters,
                           in_fmt=in_fmt, out_name=out_name)


@needs_ref
@pytest.mark.parametrize("seed", range(8000, 8016))
def test_cli_live_sharded_emul(seed):
    """The same random inputs and flag sets (downsampling, repeat gate, -D, several adapters, FASTA, .gz output ...) as a job
    of 2 or 3 rank processes: the parts concatenated in rank order, the INFO lines and the report equal the reference's."""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), "emul"], check=True)
    if seed % 2:
        reads, flags, adapters, fasta = case(seed, 60)
        cli_check.compare_live(os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"), REF, reads, flags, adapters, fasta, ranks=2 + seed % 3 % 2)
    else:
        reads, flags, adapters, in_fmt, out_name = case2(seed, 60)
        if in_fmt not in ("fq", "fa"):
            in_fmt = "fq"                          # (a sharded job takes plain text)
            out_name = out_name.replace(".fa", ".fq") if "-f" not in flags else out_name
        cli_check.compare_live(os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"), REF, reads, flags, adapters,
                               in_fmt=in_fmt, out_name=out_name, ranks=3)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8000, 8024))
def test_cli_live_sharded_gpu(seed):
    binary = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    if seed % 2:
        reads, flags, adapters, fasta = case(seed, 300)
        cli_check.compare_live(binary, REF, reads, flags, adapters, fasta, ranks=2 + seed % 3 % 2, own_args=["--devices", "0"])
    else:
        reads, flags, adapters, in_fmt, out_name = case2(seed, 300)
        if in_fmt not in ("fq", "fa"):
            in_fmt = "fq"
            out_name = out_name.replace(".fa", ".fq") if "-f" not in flags else out_name
        cli_check.compare_live(binary, REF, reads, flags, adapters, in_fmt=in_fmt, out_name=out_name, ranks=3, own_args=["--devices", "0"])


@needs_ref
@pytest.mark.parametrize("seed", range(5000, 5008))
def test_cli_live_stdout_and_adapters_only_emul(seed):
    """No -o (records on stdout, report named after the input) and -A (adapter identification only)."""
    binary = os.path.join(ROOT, "tests", "emul", "tgsfilter_emul")
    reads, flags, adapters, fasta = case(seed, 80)
    if "-r" not in flags:
        cli_check.compare_live(binary, REF, reads, flags, adapters, fasta, to_stdout=True)
    cli_check.compare_live(binary, REF, reads, flags + ["-A"], None, fasta)


def case3(seed, n):
    """Unusual but valid FASTQ texts: headers with blanks and tabs, repeated names, blank lines before records,
    Phred+64 qualities.  (Texts the reference itself cannot read -- CR LF line ends, a last line without a
    newline: it reads out of bounds there -- are accepted here and are not part of the comparison.)"""
    rng = np.random.default_rng(seed)
    reads, flags, adapters, _ = case(seed, n)
    mode = int(rng.choice([0, 1, 5, 6]))
    out = bytearray()
    for i, (name, s, q) in enumerate(reads):
        nm = name
        if mode == 0:
            nm = name + b" runid=abc\tch=5"
        if mode == 1 and i % 7 == 3:
            nm = b"dup"
        if mode == 6:
            q = bytes(min(126, c + 31) for c in q)
        rec = b"@" + nm + b"\n" + s + b"\n+\n" + q + b"\n"
        if mode == 5 and i % 9 == 4:
            rec = b"\n" + rec
        out += rec
    return bytes(out), flags, adapters


@needs_ref
@pytest.mark.parametrize("seed", range(6000, 6010))
def test_cli_live_unusual_text_emul(seed):
    raw, flags, adapters = case3(seed, 60)
    cli_check.compare_live(os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"), REF, None, flags, adapters, raw_input=raw)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6000, 6016))
def test_cli_live_unusual_text_gpu(seed):
    raw, flags, adapters = case3(seed, 300)
    cli_check.compare_live(os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter"), REF, None, flags, adapters, raw_input=raw)


def long_reads(lengths, seed=11):
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = []
    for i, L in enumerate(lengths):
        s = bytearray(acgt[rng.integers(0, 4, L)].tobytes())
        for k in range(int(rng.integers(0, 4))):
            a = synth.mutate(rng, synth.ONT_RAPID if k % 2 else synth.ONT_RAPID_RC, 0.08)
            if L > 3000:
                p = int(rng.integers(200, L - 200 - len(a)))
                s[p:p + len(a)] = a
        if i % 2 == 0 and L > 1000:
            a = synth.mutate(rng, synth.ONT_RAPID, 0.1)
            s[10:10 + len(a)] = a
        q = (np.clip(np.rint(rng.normal(14, 4, L)), 1, 50) + 33).astype(np.uint8).tobytes()
        reads.append((b"long%d" % i, bytes(s), q))
    return reads


LONG_FLAGS = [["-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0"], ["-x", "ont", "-l", "500", "-5", "3", "-3", "2", "-D"],
              ["-x", "ont", "-l", "1000", "-M", "25", "-S", "0.8", "-T", "10"], ["-x", "ont", "-l", "1000", "-p", "3", "-k", "11", "-r", "4"]]


@needs_ref
@pytest.mark.parametrize("flags", LONG_FLAGS[:2])
def test_cli_live_long_reads_emul(flags):
    """Records larger than the default text slice of a small file, tile / segment edge lengths."""
    reads = long_reads([300_000, 123_457, 6400 * 3, 6400 * 3 + 1, 6399, 6401, 102_400, 99])
    cli_check.compare_live(os.path.join(ROOT, "tests", "emul", "tgsfilter_emul"), REF, reads, flags, [synth.ONT_RAPID])


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("flags", LONG_FLAGS)
def test_cli_live_long_reads_gpu(flags):
    reads = long_reads([2_000_000, 1_234_567, 700_001, 350_000, 6400 * 3, 6400 * 3 + 1, 6399, 6401, 102_400, 99])
    cli_check.compare_live(os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter"), REF, reads, flags, [synth.ONT_RAPID])


@needs_ref
def test_cli_live_bgzipped_fastq_emul():
    """A bgzip'ed FASTQ (BGZF members, inflated side by side here) and a plain multi-member .gz."""
    import gzip
    import io
    from tests import bamio
    reads, flags, adapters, _ = case(321, 80)
    buf = io.BytesIO()
    for name, s, q in reads:
        buf.write(b"@" + name + b"\n" + s + b"\n+\n" + q + b"\n")
    text = buf.getvalue()
    binary = os.path.join(ROOT, "tests", "emul", "tgsfilter_emul")
    cli_check.compare_live(binary, REF, None, flags, adapters, in_fmt="fq.gz", raw_input=bamio.bgzf(text, block=0x3000))
    half = len(text) // 2
    cut = text.index(b"\n@", half) + 1
    two = gzip.compress(text[:cut], 1) + gzip.compress(text[cut:], 6)
    cli_check.compare_live(binary, REF, None, flags, adapters, in_fmt="fq.gz", raw_input=two)
