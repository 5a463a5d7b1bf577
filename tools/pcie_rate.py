#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (tgsf_submit: H2D of the batch, the pipeline, D2H of
the records), for the note in DESIGN.md §5 -- never the bench's `value`.  The C2-shaped batch of bench.py is
copied to pinned host memory; N host threads each own a context and submit batches back to back."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from tgsfilter_amd import abi, capi, synth  # noqa: E402


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda", 0)
    b = bench.gen_batch(torch, dev, reads, 1, 45000.0, 2_000_000, "ont")
    h = {}
    for k in ("seq", "qual"):
        t = torch.empty(b[k].numel(), dtype=torch.uint8).pin_memory()
        t.copy_(b[k])
        h[k] = t.numpy()
    off = b["h_offsets"][:-1].astype(np.uint64)
    lens = b["h_lens"].astype(np.uint32)
    p = abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_len=1000, min_q=10.0, head_trim=0,
                        tail_trim=0, max_batch_bases=b["bases"] + 64, max_batch_reads=reads, max_read_len=int(lens.max()))
    for nthreads in (1, 2, 3):
        ctxs = [capi.Context(p, 0) for _ in range(nthreads)]
        for c in ctxs:
            c.submit(h["seq"], h["qual"], off, lens)          # warm-up

        def work(c):
            for _ in range(steps):
                c.submit(h["seq"], h["qual"], off, lens)
        th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        gb = 2.0 * b["n_bytes"] * steps * nthreads / 1e9
        print("threads/contexts %d: %.1f Gbases/s  (%.1f GB/s over PCIe, pinned host buffers, %d reads / %.2f Gbases per batch)"
              % (nthreads, b["bases"] * steps * nthreads / dt / 1e9, gb / dt, reads, b["bases"] / 1e9))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
