"""Host-side glue for the parity tests: flag parsing, record formatting, stderr parsing.

This mirrors what sits on the host side of the C ABI in the reference (newSeqName,
record formatting, the stderr counter lines) so that per-read results -- from the
oracle or from the HIP library -- can be compared with whole-program goldens.
"""
from __future__ import annotations

import gzip
import json
import os
import re

import numpy as np

from tgsfilter_amd import abi, synth

ISSPACE = b" \t\n\v\f\r"


def new_seq_name(raw: bytes, number: int) -> bytes:
    """newSeqName, src/TGSFilter.cpp:1680-1701: insert ':<n>' before the first whitespace."""
    add = b":" + str(number).encode()
    for i, c in enumerate(raw):
        if bytes([c]) in ISSPACE:
            return raw[:i] + add + raw[i:]
    return raw + add


def format_fastq(reads, results, frags) -> bytes:
    """The records filter_sequence enqueues (src/TGSFilter.cpp:2011-2021), in input order."""
    out = []
    for r, (name, seq, qual) in enumerate(reads):
        rr = results[r]
        pass_num = 1
        for f in frags[rr["frag_begin"]:rr["frag_begin"] + rr["n_frags"]]:
            assert f["read"] == r
            if not (f["flags"] & abi.FF_PASS):
                continue
            nm = new_seq_name(name, pass_num) if pass_num >= 2 else name
            pass_num += 1
            s, l = int(f["start"]), int(f["len"])
            out.append(b"@" + nm + b"\n" + seq[s:s + l] + b"\n+\n" + qual[s:s + l] + b"\n")
    return b"".join(out)


def parse_flags(flags: str):
    toks = flags.split()
    kv, i = {}, 0
    while i < len(toks):
        t = toks[i].lstrip("-")
        if t in ("D", "qc", "A", "F", "f"):
            kv[t] = True
            i += 1
        else:
            kv[t] = toks[i + 1]
            i += 2
    return kv


def parse_stderr(text: str):
    """Pull the resolved parameters and the 17 DropInfo counters out of the reference's stderr."""
    d = {"adapters": []}
    pats = {
        "qtype": r"base quality scoring: Phred(\d+)",
        "head_trim": r"trim 5' end length: (-?\d+)",
        "tail_trim": r"trim 3' end length: (-?\d+)",
        "min_q": r"min Phred average quality score: ([\d.]+)",
        "mid_sim": r"min similarity for middle adapter: ([\d.]+)",
        "end_sim": r"min similarity for end adapter: ([\d.]+)",
    }
    for k, p in pats.items():
        m = re.search(p, text)
        if m:
            d[k] = float(m.group(1)) if "." in m.group(1) or k in ("min_q",) else int(m.group(1))
    for m in re.finditer(r"input adapter \d+ :(\S+)", text):
        d["adapters"].append(m.group(1).encode())
    m5 = re.search(r"INFO: 5' adapter: (\S*)", text)
    m3 = re.search(r"INFO: 3' adapter: (\S*)", text)
    d["adapter5p"] = m5.group(1).encode() if m5 else None
    d["adapter3p"] = m3.group(1).encode() if m3 else None
    m = re.search(r"set (?:PacBio blunt|NanoPore rapid) adapter to trim: (\S+)", text)
    d["default_adapter"] = m.group(1).encode() if m else None
    drop = [None] * 17
    m = re.search(r"INFO: (\d+) reads with a total of (\d+) bases were input", text)
    if m:
        d["raw_reads"], d["raw_bases"] = int(m.group(1)), int(m.group(2))
    m = re.search(r"INFO: (\d+) reads were discarded with (\d+) bases due to low quality\.", text)
    if m:
        drop[0], drop[1] = int(m.group(1)), int(m.group(2))
    for idx, tail in [(2, "at 5', 3' and middle"), (3, "at 5' and middle"), (4, "at 3' and middle"),
                      (5, "at 5' and 3' end"), (6, "only have adapter at middle"),
                      (7, "only have adapter at 5' end"), (8, "only have adapter at 3' end"),
                      (9, "didn't have any adapter")]:
        m = re.search(r"INFO: (\d+) reads (?:have adapter |)" + re.escape(tail), text)
        if m:
            drop[idx] = int(m.group(1))
    m = re.search(r"INFO: (\d+) bases were trimmed", text)
    if m:
        drop[10] = int(m.group(1))
    m = re.search(r"INFO: (\d+) reads were discarded with (\d+) bases due to the short length", text)
    if m:
        drop[11], drop[12] = int(m.group(1)), int(m.group(2))
    m = re.search(r"INFO: (\d+) reads were discarded with (\d+) bases due to low quality after split", text)
    if m:
        drop[13], drop[14] = int(m.group(1)), int(m.group(2))
    m = re.search(r"INFO: (\d+) reads with a total of (\d+) bases after filtering", text)
    if m:
        d["clean_reads"], d["clean_bases"] = int(m.group(1)), int(m.group(2))
    d["drop"] = drop
    return d


class GoldenCase:
    """One tests/golden/<name>.* fixture: inputs, the reference's outputs and the resolved Params."""

    def __init__(self, golden_dir: str, name: str):
        self.name = name
        self.cmd = json.load(open(os.path.join(golden_dir, name + ".cmd.json")))
        self.reads = synth.read_fastq(os.path.join(golden_dir, name + ".in.fq.gz"))
        self.ref_out = gzip.open(os.path.join(golden_dir, name + ".out.fq.gz"), "rb").read()
        self.stderr = open(os.path.join(golden_dir, name + ".stderr.txt")).read()
        self.html = json.load(open(os.path.join(golden_dir, name + ".html.json")))
        self.info = parse_stderr(self.stderr)
        self.flags = parse_flags(self.cmd["flags"])

    def params(self, **extra) -> abi.Params:
        """Params as the reference resolved them (explicit flags + what its stderr reports)."""
        f, info = self.flags, self.info
        if f.get("qc"):
            return abi.make_params("ont", only_qc=True, filter=False, qtype=info.get("qtype", 33),
                                   bc_len=int(f.get("e", 150)), **extra)
        rt = f["x"].lower()
        rt = "hifi" if rt == "ccs" else rt
        if info["adapters"]:
            ads = info["adapters"]
        else:
            ads = []
            for a in (info["adapter5p"], info["adapter3p"]):
                if a:
                    for x in (a, synth.revcomp(a)):
                        if x not in ads:
                            ads.append(x)
            if not ads and info["default_adapter"]:
                a = info["default_adapter"]
                ads = [a, synth.revcomp(a)]
        return abi.make_params(
            rt, adapters=ads, min_len=int(f.get("l", 1000)), max_len=int(f.get("L", 2147483647)),
            min_q=info["min_q"], max_q=float(f.get("Q", 255)), bc_len=int(f.get("e", 150)),
            head_trim=info["head_trim"], tail_trim=info["tail_trim"], end_len=int(f.get("E", 150)),
            end_match_len=int(f.get("m", 4)), mid_match_len=int(f.get("M", 35)),
            extra_len=int(f.get("T", 50)), end_sim=float(f["s"]) if "s" in f else None,
            mid_sim=float(f["S"]) if "S" in f else None, discard=bool(f.get("D")),
            qtype=info.get("qtype", 33), **extra)


GOLDEN_CASES = ["ont_zoo", "ont_trim", "ont_discard", "hifi_zoo", "long_adapter", "ont_auto",
                "hifi_auto", "qc_only", "ont_m1", "huge_adapter", "ont_phred64", "hifi_phred64_auto", "ont_e1300", "wide_adapter", "ont_high_qual", "giant_adapter"]
# whole-program cases that involve the host-side second pass (checked through the CLI only)
# ... or a non-FASTQ input format (SAM / unaligned BAM decoded by the host; FASTA = records without qualities)
GOLDEN_CLI_ONLY = ["ont_repeat", "repeat_k15", "repeat_k21", "repeat_k32", "repeat_k32b", "down_gd", "down_r", "down_R", "down_F", "hifi_bam", "ont_sam", "hifi_bam_auto",
                   "ont_fasta", "hifi_fasta_auto", "fasta_down"]
