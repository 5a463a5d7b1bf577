"""Seeded parameter / input fuzzing shared by the emulation (CPU) and GPU suites: random command-line-level
parameters inside the supported domain, random read sets, full comparison with the oracle."""
from __future__ import annotations

import os

import numpy as np

from tests import parity
from tgsfilter_amd import abi, capi, synth

LIB_ADAPTERS = [synth.ONT_RAPID, synth.ONT_RAPID_RC, synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC,
                b"AATGTACTTCGTTCAGTTACGTATTGCT", b"AGCAATACGTAACTGAACGAAGTACATT",
                b"GGCGTCTGCTTGGGTGTTTAACCTTTTTGTCAGAGAGGTTCCAAGTCAGAGAGGTTCCT",
                b"CTTGCGGGCGGCGGACTCTCCTCTGAAGATAGAGCGACAGGCAAG",
                # beyond the library: 3- and 4-word adapters (-a accepts any length; the reference's edlib is multi-block)
                bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(150).integers(0, 4, 150)]),
                bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(241).integers(0, 4, 241)])]


# round 3's additions to the domain (TGSF_FUZZ_WIDE=1; the seeds of earlier campaigns keep their meaning without it): 22-bp
# adapters (the one-dword scan column, also mixed with longer ones), -e beyond one slab of the end-table kernel, and now and
# then a candidate pool of a few slots, so that the count-and-rescan path of tgsf_wait runs on ordinary inputs too
WIDE_ADAPTERS = [b"GCAATACGTAACTGAACGAAGT", b"ACTTCGTTCAGTTACGTATTGC", b"ACGTAACTGAACGAAGTACAGG", b"TTTTTTTTCCTGTACTTCGTTCAGTTACGT"]


def random_case(seed: int, n_reads: int):
    rng = np.random.default_rng(seed)
    wide = os.environ.get("TGSF_FUZZ_WIDE") == "1"
    kind = "ont" if rng.random() < 0.6 else "hifi"
    na = int(rng.choice([1, 2, 2, 3, 4]))
    pool = LIB_ADAPTERS + (WIDE_ADAPTERS if wide else [])
    ads = [pool[i] for i in rng.choice(len(pool), na, replace=False)]
    planted = ads[0] if rng.random() < 0.8 else None
    mean_len = float(rng.choice([800, 2500, 6000]))
    if os.environ.get("TGSF_FUZZ_MEAN_LEN"):                          # (a campaign can ask for long reads: windows of the repeat gate, many scan blocks)
        mean_len = float(os.environ["TGSF_FUZZ_MEAN_LEN"])
    reads = synth.make_reads(int(rng.integers(1, 1 << 30)), n_reads, kind, mean_len=mean_len,
                             zoo=bool(rng.random() < 0.7), pmid=float(rng.choice([0.0, 0.05, 0.3])),
                             **({"adapter": planted} if planted else {}))
    end_len = int(rng.choice([150, 60, 300]))
    kw = dict(
        adapters=ads,
        min_len=int(rng.choice([100, 500, 1000, 2000])),
        max_len=int(rng.choice([2147483647, 2147483647, 5000])),
        min_q=float(rng.choice([0.0, 7.0, 10.0, 20.0])),
        max_q=float(rng.choice([255.0, 255.0, 25.0])),
        bc_len=int(rng.choice([150, 64, 1, 200])),
        head_trim=int(rng.choice([0, 0, 5, 40])),
        tail_trim=int(rng.choice([0, 0, 7, 33])),
        end_len=end_len,
        end_match_len=int(rng.choice([4, 8, 15, 1])),           # 1: edlib's k >= Q corner (-m 1 / -M 1)
        mid_match_len=int(rng.choice([35, 20, 25, 35, 1])),
        extra_len=int(rng.choice([50, 0, 10])),
        end_sim=float(rng.choice([0.75, 0.8, 0.9])),
        mid_sim=float(rng.choice([0.9, 0.95, 0.8])),
        discard=bool(rng.random() < 0.25),
        filter=bool(rng.random() < 0.9),
        qtype=33,
    )
    if rng.random() < float(os.environ.get("TGSF_FUZZ_GATE_P", "0.2")):      # (a campaign can ask for the repeat gate more often)
        kw.update(min_repeat=int(rng.choice([50, 400, 1, 5])), kmer=int(rng.choice([7, 11, 12, 14, 20, 32, 9, 13, 15, 16, 17, 31])))
    if rng.random() < 0.15:
        kw.update(no_qual=True)
    if wide:
        if rng.random() < 0.3:
            kw.update(bc_len=int(rng.choice([513, 700, 1300])))
        if rng.random() < 0.5:
            kw.update(mid_match_len=int(rng.choice([12, 16, 18, 22])))       # within reach of the short adapters
        kw["_pool_cap"] = int(rng.choice([0, 0, 0, 3, 40]))                   # 0: the context's own sizing
    if os.environ.get("TGSF_FUZZ_TRIMS") == "1":
        # (round 6, the split-bin by-product: trims on every seam of a bin -- drawn from a stream of their own, so that the
        # cases of the other campaigns and of the recorded findings stay what they were)
        r2 = np.random.default_rng(seed ^ 0x5EED7A11)
        kw.update(head_trim=int(r2.choice([0, 1, 3, 4, 79, 99, 100, 101, 137, 250, 6400, 6401])), tail_trim=int(r2.choice([0, 0, 1, 8, 99, 100, 120])),
                  filter=True, min_len=int(r2.choice([100, 500, 1000])))
        if r2.random() < 0.5:                                             # (the by-product beside the repeat gate: round 6)
            kw.pop("min_repeat", None); kw.pop("kmer", None)
        kw["_force_by_product"] = bool(r2.random() < 0.7)
    return kind, reads, kw


def run_case(lib_path, seed: int, n_reads: int):
    kind, reads, kw = random_case(seed, n_reads)
    pool_cap = kw.pop("_pool_cap", 0)
    force_bp = kw.pop("_force_by_product", False)
    outer_ct = os.environ.get("TGSF_CLEAN_TABLES")
    if force_bp:
        os.environ["TGSF_CLEAN_TABLES"] = "byproduct"
    p = parity.sized(abi.make_params(kind, **kw), reads)
    outer = os.environ.get("TGSF_POOL_CAP")                 # (a campaign may set one for every case: keep it)
    if pool_cap:
        os.environ["TGSF_POOL_CAP"] = str(pool_cap)
    try:
        ctx = capi.Context(p, 0, lib_path)
    finally:
        if outer is None:
            os.environ.pop("TGSF_POOL_CAP", None)
        else:
            os.environ["TGSF_POOL_CAP"] = outer
        if force_bp:
            if outer_ct is None:
                os.environ.pop("TGSF_CLEAN_TABLES", None)
            else:
                os.environ["TGSF_CLEAN_TABLES"] = outer_ct
    try:
        parity.compare_batch(ctx, p, reads, align=int(np.random.default_rng(seed).choice([1, 16])),
                             explicit_lengths=True)
    except AssertionError as e:
        raise AssertionError("fuzz seed %d (%s, %s): %s" % (seed, kind, {k: v for k, v in kw.items() if k != "adapters"}, e))
    finally:
        ctx.close()
