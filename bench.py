#!/usr/bin/env python3
"""bench.py -- filtered Gbases/s of the per-read filtering hot path on N x MI355X.

Contract (see the driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched by torch.distributed.run, one rank per GPU.  A step = one pass of the hot path over one
batch of synthetic ONT reads (config C2 of BASELINE.json: lognormal lengths, mean 45 kb,
`-x ont -l 1000 -q 10`, ONT rapid adapter + reverse complement) that is already resident in HBM.
Reads shard across ranks with no data-path collective; the only exchange is one all-reduce
(RCCL) of the tally vector at the end of the job, inside the timed region.

Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel (middle
adapter scan, timed with HIP events on the launch stream inside the library) and -- at N = 1 --
the reference's own CPU path (oracle/_ref/tgsfilter_ref -t <cores>) timed on this box's host cores
on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def gen_batch(torch, device, n_reads, seed, mean_len, max_len, workload="ont"):
    """Synthetic C2 (ont) / C3 (hifi) batch built directly in HBM (generation is outside every timed region)."""
    from tgsfilter_amd import synth
    rng = np.random.default_rng(seed)
    hifi = workload == "hifi"
    if hifi:
        lens = np.minimum(synth.hifi_lengths(rng, n_reads, mean_len, sd=mean_len / 6.0), max_len).astype(np.int64)
    else:
        lens = np.minimum(synth.ont_lengths(rng, n_reads, mean_len), max_len).astype(np.int64)
    padded = (lens + 15) // 16 * 16
    offsets = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(padded, out=offsets[1:])
    total = int(offsets[-1])
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    seq = torch.empty(total, dtype=torch.uint8, device=device)
    qual = torch.empty(total, dtype=torch.uint8, device=device)
    mq_set = np.array([30], dtype=np.float32) if hifi else np.array([7, 9, 12, 14, 18], dtype=np.float32)
    mq = torch.from_numpy(rng.choice(mq_set, n_reads)).to(device)
    per_base_mq = torch.repeat_interleave(mq.to(torch.float16), torch.from_numpy(padded).to(device))
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    CH = 1 << 26
    for s in range(0, total, CH):
        e = min(total, s + CH)
        codes = torch.randint(0, 4, (e - s,), device=device, generator=g)
        seq[s:e] = lut[codes]
        q = torch.randn(e - s, device=device, generator=g) * (6.0 if hifi else 4.0) + per_base_mq[s:e].float()
        qual[s:e] = (q.round().clamp_(2 if hifi else 1, 60 if hifi else 50) + 33).to(torch.uint8)
    del per_base_mq
    # rapid adapter at the 5' end of 80 % of reads (0-30 random bases before it, 10 % errors); 0.03 % middle
    idx, val = [], []
    for i in range(n_reads):
        L = int(lens[i])
        if hifi:
            # blunt adapter 5' in 0.27 %, 3' in 0.26 %, middle in 0.002 % of reads, 3 % errors (README ratios)
            u = rng.random()
            if u < 0.0027 + 0.0026 + 0.00002 and L > 2000:
                a = synth.mutate(rng, synth.PACBIO_BLUNT, 0.03)
                p = 0 if u < 0.0027 else (L - len(a) if u < 0.0053 else int(rng.integers(300, L - 300 - len(a))))
                idx.append(np.arange(len(a), dtype=np.int64) + offsets[i] + p)
                val.append(np.frombuffer(a, dtype=np.uint8))
            continue
        if rng.random() < 0.80:
            a = synth.mutate(rng, synth.ONT_RAPID, 0.10)
            pre = int(rng.integers(0, 31))
            if pre + len(a) < L:
                idx.append(np.arange(len(a), dtype=np.int64) + offsets[i] + pre)
                val.append(np.frombuffer(a, dtype=np.uint8))
        if rng.random() < 0.0003 and L > 2000:
            a = synth.mutate(rng, synth.ONT_RAPID if rng.random() < 0.5 else synth.ONT_RAPID_RC, 0.05)
            p = int(rng.integers(300, L - 300 - len(a)))
            idx.append(np.arange(len(a), dtype=np.int64) + offsets[i] + p)
            val.append(np.frombuffer(a, dtype=np.uint8))
    if idx:
        seq[torch.from_numpy(np.concatenate(idx)).to(device)] = torch.from_numpy(np.concatenate(val).copy()).to(device)
    return dict(seq=seq, qual=qual, offsets=torch.from_numpy(offsets[:-1].astype(np.uint64).view(np.int64).copy()).to(device),
                lengths=torch.from_numpy(lens.astype(np.uint32).view(np.int32).copy()).to(device),
                n=n_reads, n_bytes=total, bases=int(lens.sum()), h_lens=lens, h_offsets=offsets)


def cpu_baseline(torch, batch, flags, adapter_fa, sample_reads):
    """The reference's own CPU path on this box's host cores, on a bounded sample of the batch."""
    ref = os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")
    n = min(sample_reads, batch["n"])
    lens, offs = batch["h_lens"][:n], batch["h_offsets"][:n + 1]
    end = int(offs[n])
    seq = batch["seq"][:end].cpu().numpy()
    qual = batch["qual"][:end].cpu().numpy()
    bases = int(lens.sum())
    cores = max(1, min((os.cpu_count() or 2) - 1, 32))     # the reference clamps -t to min(hw-1, 32)
    if os.path.exists(ref):
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        with tempfile.TemporaryDirectory(dir=tmpdir) as td:
            fq = os.path.join(td, "sample.fq")
            with open(fq, "wb") as f:
                for i in range(n):
                    o, L = int(offs[i]), int(lens[i])
                    f.write(b"@r%d\n" % i)
                    f.write(seq[o:o + L].tobytes())
                    f.write(b"\n+\n")
                    f.write(qual[o:o + L].tobytes())
                    f.write(b"\n")
            fa = os.path.join(td, "adapters.fa")
            open(fa, "wb").write(adapter_fa)
            cmd = [ref, "-i", fq, "-o", os.path.join(td, "out.fq"), "-a", fa, "-t", str(cores)] + flags.split()
            t0 = time.perf_counter()
            p = subprocess.run(cmd, capture_output=True)
            dt = time.perf_counter() - t0
        if p.returncode == 0:
            return {"value": bases / dt / 1e9, "unit": "Gbases/s", "cores": cores, "kind": "reference",
                    "sample": "%d reads / %.1f Mbases of the step-0 batch as uncompressed FASTQ on tmpfs, "
                              "tgsfilter_ref -t %d %s, wall %.2f s" % (n, bases / 1e6, cores, flags, dt)}
    # no reference binary on this box: the single-threaded C restatement on a smaller sample
    from oracle import orc
    from tgsfilter_amd import abi, synth
    m = min(n, 64)
    end = int(offs[m])
    p = abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=10.0,
                        max_read_len=int(lens[:m].max()))
    t0 = time.perf_counter()
    orc.filter_batch(p, seq[:end], qual[:end], offs[:m].astype(np.uint64), lens[:m].astype(np.uint32))
    dt = time.perf_counter() - t0
    b = int(lens[:m].sum())
    return {"value": b / dt / 1e9, "unit": "Gbases/s", "cores": 1, "kind": "port",
            "sample": "%d reads / %.1f Mbases, oracle/liborc.so (plain DP restatement), wall %.2f s" % (m, b / 1e6, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=131072, help="reads per step per GPU")
    ap.add_argument("--mean-len", type=float, default=None)
    ap.add_argument("--min-repeat", type=int, default=0, help="-p of config C5 (with -k 11), for information")
    ap.add_argument("--workload", choices=["ont", "hifi"], default="ont",
                    help="ont = config C2 (the headline line); hifi = config C3 shape, for information")
    ap.add_argument("--max-len", type=int, default=2_000_000)
    ap.add_argument("--cpu-sample-reads", type=int, default=24000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for validation)")
    ap.add_argument("--share-gpu", action="store_true", help="validation on a 1-GPU box: every rank uses device 0")
    ap.add_argument("--streams", type=int, default=3,
                    help="batches in flight per GPU (one context + one HIP stream each); >1 overlaps the "
                         "HBM-bound stats kernels of one batch with the VALU-bound adapter scan of another")
    args = ap.parse_args()
    # stdout carries exactly one line, the result: whatever libraries print there meanwhile goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from tgsfilter_amd import abi, capi, synth
    from tgsfilter_amd import dist as tdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.share_gpu:                      # validation on a 1-GPU box: all ranks on device 0, gloo for the exchange
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    xdev = device if args.backend == "nccl" else torch.device("cpu")       # where the exchanged tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    hifi = args.workload == "hifi"
    if args.mean_len is None:
        args.mean_len = 18000.0 if hifi else 45000.0
    flags = "-x hifi -l 1000 -q 20 -5 0 -3 0" if hifi else "-x ont -l 1000 -q 10 -5 0 -3 0"
    wl_adapters = [synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC] if hifi else [synth.ONT_RAPID, synth.ONT_RAPID_RC]
    batches = [gen_batch(torch, device, args.reads, 1000 * rank + b + 1, args.mean_len, args.max_len, args.workload)
               for b in range(2)]
    max_bases = max(b["bases"] for b in batches)
    max_len = max(int(b["h_lens"].max()) for b in batches)
    if world > 1:
        # the tally vector is sized by max_read_len (rows of the per-position tables): every rank must build
        # the same layout, or the one all-reduce of the job would mix up words
        mm = torch.tensor([max_bases, max_len], dtype=torch.int64, device=xdev)
        dist.all_reduce(mm, op=dist.ReduceOp.MAX)
        max_bases, max_len = int(mm[0].item()), int(mm[1].item())
    p = abi.make_params(args.workload, adapters=wl_adapters, min_len=1000, min_q=20.0 if hifi else 10.0,
                        head_trim=0, tail_trim=0, max_batch_bases=max_bases + 64, max_batch_reads=args.reads,
                        max_read_len=max_len, min_repeat=args.min_repeat, kmer=11)
    NS = max(1, args.streams)
    ctxs = [capi.Context(p, local_rank) for _ in range(NS)]
    ctx = ctxs[0]
    fcap = max_bases // 1000 + args.reads + 16
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=device) for _ in range(NS - 1)]
    outs = [dict(reads=torch.empty(args.reads * 32, dtype=torch.uint8, device=device),
                 frags=torch.empty(fcap * 24, dtype=torch.uint8, device=device),
                 nfr=torch.zeros(4, dtype=torch.int32, device=device),
                 h_reads=torch.empty(args.reads * 32, dtype=torch.uint8).pin_memory()) for _ in range(NS)]

    def step(i):
        b = batches[i % 2]
        k = i % NS if NS > 1 else 0
        o = outs[k]
        with torch.cuda.stream(streams[k]):
            ctxs[k].submit_device(b["seq"].data_ptr(), b["qual"].data_ptr(), b["offsets"].data_ptr(),
                                  b["lengths"].data_ptr(), b["n"], b["n_bytes"], o["reads"].data_ptr(),
                                  o["frags"].data_ptr(), fcap, o["nfr"].data_ptr(), streams[k].cuda_stream)
            o["h_reads"].copy_(o["reads"], non_blocking=True)   # the per-read records go back to the host every step
        return b["bases"], b["n"]

    ctr_words = ctx.ctr_words

    def all_wait():
        for c in ctxs:
            c.wait()

    def all_counters():
        return tdist.merge_counters([c.counters() for c in ctxs])

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    all_wait()
    for c in ctxs:
        c.reset_counters()
        c.profile(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bases = reads = 0
    for i in range(args.steps):
        b, n = step(i)
        bases += b
        reads += n
    if world > 1:
        # the job's only exchange: sum the tally vector over ranks (the 4 "rows used" words are maxima)
        all_wait()
        total_ctr = tdist.allreduce_counters(all_counters(), device=xdev if args.backend == "nccl" else None)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    all_wait()
    if world == 1:
        total_ctr = all_counters()

    # max over ranks of the elapsed time; sum of the bases
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        bb = torch.tensor([bases, reads], dtype=torch.int64, device=xdev)
        dist.all_reduce(bb, op=dist.ReduceOp.SUM)
        bases_all, reads_all = int(bb[0].item()), int(bb[1].item())
    else:
        bases_all, reads_all = bases, reads

    # sanity: every read was classified exactly once (low-Q, or one of the 8 adapter classes)
    drop = total_ctr[:17]
    assert int(drop[0]) + int(drop[2:10].sum()) == reads_all, (drop, reads_all)

    def harvest():
        st, nbat = {}, 0
        for c in ctxs:
            st_c, nb_c = c.stage_times()
            nbat += nb_c
            for k, v in st_c.items():
                st[k] = st.get(k, 0.0) + v
        return {k: v / max(nbat, 1) for k, v in st.items() if v > 0}, nbat

    # kernel durations inside the timed region (with --streams > 1 kernels of different batches
    # share the GPU, so a kernel's elapsed time is longer than its cost) ...
    timed_stage_ms, _ = harvest()
    # ... and the same kernels with the GPU to themselves: a few more steps on ONE stream, timed with
    # the same HIP events on the launch stream.  The roofline figures use these exclusive durations.
    for c in ctxs:
        c.profile(False)
    NS_saved, nprof = NS, 4
    ctxs[0].profile(True)
    NS = 1
    for i in range(nprof):
        step(i)
    torch.cuda.synchronize()
    ctxs[0].wait()
    excl_stage_ms, _ = harvest()
    NS = NS_saved
    dom = "mid_scan"
    t_dom = excl_stage_ms[dom] / 1e3                       # seconds per launch of the dominant kernel
    # stages 'end_tables_raw' and 'end_windows' run on the library's auxiliary stream beside 'mid_scan'
    t_all = sum(v for k, v in excl_stage_ms.items() if k not in ("end_tables_raw", "end_windows")) / 1e3
    alg_bytes = 2.0 * (bases / args.steps) + 32.0 * (reads / args.steps)    # 2 B/base + 32 B/read
    achieved = alg_bytes / t_dom / 1e9 if t_dom > 0 else 0.0
    traffic = None
    valu = None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        tr = json.load(open(tj))
        if tr.get("reads_per_step") == args.reads and args.workload == "ont":
            traffic = tr.get("mid_scan_hbm_bytes_per_launch")
            vi, va = tr.get("mid_scan_valu_insts_per_launch"), tr.get("valu_insts_per_batch_all_kernels")
            if vi and va and t_dom > 0:
                # a wave64 VALU instruction holds its SIMD for 4 cycles; 256 CUs x 4 SIMDs
                valu = {"kernel_valu_insts_per_launch": vi, "kernel_issue_cycles_per_simd": vi * 4 / 1024,
                        "kernel_min_clock_ghz_if_valu_only": vi * 4 / 1024 / t_dom / 1e9,
                        "pipeline_valu_insts_per_batch": va, "pipeline_issue_cycles_per_simd": va * 4 / 1024,
                        "note": "SQ_INSTS_VALU from profiles/ (PMC pass of the same command): the dominant kernel "
                                "issues VALU instructions back to back for its whole duration -- the pipeline is "
                                "VALU-issue bound, the HBM fraction above is what that leaves"}

    out = {
        "metric": "filtered Gbases/sec (end-to-end, excl. gzip I/O)",
        "value": bases_all / dt / 1e9,
        "unit": "Gbases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": ("C3 shape (BASELINE.json configs[2], for information): synthetic HiFi reads, N(%.0f, /6) bp, Q~N(30,6), "
                         "%s, adapters PacBio blunt + reverse complement; %d reads (%.2f Gbases) per step per GPU, "
                         "inputs resident in HBM" % (args.mean_len, flags, args.reads, bases / args.steps / 1e9)) if hifi else
                        ("C2 (BASELINE.json configs[1]): synthetic ONT reads, lognormal lengths mean %.0f bp, "
                         "%s, adapters ONT rapid + reverse complement; %d reads (%.2f Gbases) per step per GPU, "
                         "inputs resident in HBM; the 4M-read job is %d such steps"
                         % (args.mean_len, flags, args.reads, bases / args.steps / 1e9,
                            int(np.ceil(4_000_000 / args.reads)))),
            "reads_per_step_per_gpu": args.reads,
            "parallelism": "reads sharded over %d GPU(s), one all-reduce of the tallies" % world,
            "batches_in_flight_per_gpu": NS,
        },
        "roofline": {
            "bound": "hbm", "kernel": "k_mid_scan1<2> (Myers infix scan, stage 'mid_scan')",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "valu_issue": valu,
            "measured": "HIP events on the launch stream around every stage (inside libtgsf); kernel durations "
                        "of %d single-stream steps run right after the timed region (the GPU is not shared with "
                        "another batch); 'timed_region_stage_ms' are the same events inside the timed region with "
                        "%d batches in flight" % (nprof, NS),
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": excl_stage_ms[dom],
            "pipeline_achieved": alg_bytes / t_all / 1e9 if t_all > 0 else 0.0,
            "pipeline_frac": (alg_bytes / t_all / 1e9 / HBM_PEAK_GBS) if t_all > 0 else 0.0,
            "whole_job_frac": (2.0 * bases_all / world + 32.0 * reads_all / world) / dt / 1e9 / HBM_PEAK_GBS,
            "stage_ms_per_step": excl_stage_ms,
            "timed_region_stage_ms": timed_stage_ms,
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, batches[0], flags,
                                               b">ad\n" + wl_adapters[0] + b"\n", args.cpu_sample_reads)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
