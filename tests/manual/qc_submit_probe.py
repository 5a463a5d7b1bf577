#!/usr/bin/env python3
"""How long one tgsf_submit of a large QC-only batch from ordinary host memory takes (the downsampling run's second pass)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tgsfilter_amd import abi, capi, synth
reads = synth.make_reads(3, 20000, "ont", mean_len=45000)
seq, qual, offsets, lengths = synth.pack(reads)
print("%d reads, %.2f Gbases" % (len(reads), lengths.sum() / 1e9))
for only_qc, mb in ((True, 1 << 30), (True, 256 << 20), (False, 256 << 20)):
    p = abi.make_params("ont", adapters=[] if only_qc else [synth.ONT_RAPID], min_q=10.0, filter=not only_qc, only_qc=only_qc)
    p.max_batch_bases = mb; p.max_batch_reads = 1 << 16; p.max_read_len = int(lengths.max())
    t0 = time.perf_counter(); ctx = capi.Context(p, 0); t1 = time.perf_counter()
    # batches of at most mb bytes
    i = 0; n = len(lengths); times = []; nb = 0
    offs = offsets[:-1]
    while i < n:
        j = i
        while j < n and offsets[j + 1] - offsets[i] <= mb - (1 << 20): j += 1
        sub_off = (offs[i:j] - offsets[i]).astype(np.uint64); sub_len = lengths[i:j]
        a, b = int(offsets[i]), int(offsets[j])
        s0 = time.perf_counter()
        ctx.submit(seq[a:b + 64], qual[a:b + 64], sub_off, sub_len)
        times.append(time.perf_counter() - s0); nb += b - a
        i = j
    print("only_qc %s, batches of %4d MB: create %.3f s; %d submits, %.3f s in all (%.1f GB/s of seq+qual); first %.3f, rest mean %.3f" % (
        only_qc, mb >> 20, t1 - t0, len(times), sum(times), 2 * nb / sum(times) / 1e9, times[0], np.mean(times[1:]) if len(times) > 1 else 0))
    ctx.close()
