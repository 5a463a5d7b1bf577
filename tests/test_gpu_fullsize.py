"""-m gpu: BASELINE.json-sized batches (C2 shape: 131 072 ONT reads of 45 kb = the batch bench.py times, resident in HBM)
checked through size-independent properties -- the oracle cannot run that much -- plus an oracle spot check
of a slice of the same batch:
  * the result does not depend on how the middle scan is cut into segments (TGSF_SEG_COLS),
  * nor on how the clean tables are tallied (TGSF_CLEAN_TABLES),
  * permuting the reads of a batch permutes the per-read results and leaves the tallies unchanged,
  * every read lands in exactly one class, bases are conserved (kept + trimmed + dropped = input)."""
import os

import numpy as np
import pytest

from tgsfilter_amd import abi, capi, synth

pytestmark = pytest.mark.gpu
N_READS = 131072        # the batch bench.py times (5.9 Gbases)


@pytest.fixture(scope="module")
def batch():
    import torch
    import bench
    dev = torch.device("cuda", 0)
    b = bench.gen_batch(torch, dev, N_READS, 7, 45000.0, 2_000_000, "ont")
    b["torch"], b["dev"] = torch, dev
    return b


def run(batch, env=None, perm=None):
    torch, dev = batch["torch"], batch["dev"]
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        n = batch["n"]
        p = abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_len=1000, min_q=10.0,
                            head_trim=0, tail_trim=0, max_batch_bases=batch["bases"] + 64, max_batch_reads=n,
                            max_read_len=int(batch["h_lens"].max()))
        ctx = capi.Context(p, 0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    off, ln = batch["offsets"], batch["lengths"]
    if perm is not None:
        pt = torch.from_numpy(perm).to(dev)
        off, ln = off[pt].contiguous(), ln[pt].contiguous()
    fcap = batch["bases"] // 1000 + n + 16
    d_reads = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_frags = torch.zeros(fcap * 24, dtype=torch.uint8, device=dev)
    d_nf = torch.zeros(4, dtype=torch.int32, device=dev)
    ctx.submit_device(batch["seq"].data_ptr(), batch["qual"].data_ptr(), off.data_ptr(), ln.data_ptr(), n,
                      batch["n_bytes"], d_reads.data_ptr(), d_frags.data_ptr(), fcap, d_nf.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    ctx.wait()
    nf = int(d_nf[0].item())
    reads = d_reads.cpu().numpy().view(abi.READ_RESULT_DTYPE)
    frags = d_frags.cpu().numpy().view(abi.FRAGMENT_DTYPE)[:nf].copy()
    ctr = ctx.counters()
    ctx.close()
    return reads, frags, ctr


def same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_full_size_segmenting_invariance(batch):
    ref = run(batch)
    for sc in ("512", "4096"):
        assert same(ref, run(batch, {"TGSF_SEG_COLS": sc})), sc


def test_full_size_clean_table_strategy_invariance(batch):
    a = run(batch, {"TGSF_CLEAN_TABLES": "direct"})
    b = run(batch, {"TGSF_CLEAN_TABLES": "difference"})
    assert same(a, b)


def test_full_size_permutation(batch):
    reads, frags, ctr = run(batch)
    perm = np.random.default_rng(3).permutation(batch["n"]).astype(np.int64)
    preads, pfrags, pctr = run(batch, perm=perm)
    assert np.array_equal(ctr, pctr)
    for name in ("sum_q", "flags", "n_frags", "trimmed"):
        assert np.array_equal(preads[name], reads[name][perm]), name
    # fragments of read perm[i] in the permuted run == fragments of that read in the original run
    for i in np.random.default_rng(4).integers(0, batch["n"], 2000):
        a = pfrags[preads["frag_begin"][i]:preads["frag_begin"][i] + preads["n_frags"][i]]
        r = perm[i]
        b = frags[reads["frag_begin"][r]:reads["frag_begin"][r] + reads["n_frags"][r]]
        assert np.array_equal(a[["start", "len", "flags", "sum_q"]], b[["start", "len", "flags", "sum_q"]])


def test_full_size_conservation_and_oracle_slice(batch):
    from oracle import orc
    reads, frags, ctr = run(batch)
    n, lens = batch["n"], batch["h_lens"].astype(np.int64)
    drop = ctr[abi.CTR_DROPINFO:abi.CTR_DROPINFO + 17].astype(np.int64)
    assert drop[0] + drop[2:10].sum() == n                       # one class per read
    kept = frags["len"][(frags["flags"] & abi.FF_PASS) != 0].astype(np.int64).sum()
    # input bases = low-Q reads + trimmed + too-short/too-long fragments + low-Q fragments + kept
    assert lens.sum() == drop[1] + drop[10] + drop[12] + drop[14] + kept
    assert int(ctr[abi.CTR_RAW_DIFFQ:abi.CTR_RAW_DIFFQ + 256].sum()) == lens.sum()
    # oracle on 500 random reads of the batch: per-read records and fragments must agree
    import bench
    kw = dict(adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_len=1000, min_q=10.0, head_trim=0, tail_trim=0)
    n_chk, _ = bench.oracle_slice_check(batch["torch"], batch, kw, "ont", reads, frags, m=500, seed=5)
    assert n_chk == 500


def test_full_size_through_the_position_ordered_rescan(batch, monkeypatch, capfd):
    """A BASELINE-sized batch (5.9 Gbases, ~6 M scan segments) whose candidate pool holds 1 000 slots: tgsf_wait counts, sizes and
    rescans (millions of (segment, adapter) cells through the prefix sum), and the records and tallies are those of the one-pass run."""
    ref_reads, ref_frags, ref_ctr = run(batch)
    monkeypatch.setenv("TGSF_POOL_CAP", "1000")
    monkeypatch.setenv("TGSF_TRACE_POOL", "1")
    reads, frags, ctr = run(batch)
    assert "scanning again in position order" in capfd.readouterr().err
    assert np.array_equal(reads, ref_reads) and np.array_equal(frags, ref_frags) and np.array_equal(ctr, ref_ctr)


def test_full_size_concurrent_contexts(batch):
    """Three contexts on three streams filter the same batch at the same time, twice each (what bench.py does):
    every one must produce the single-context result; tallies double."""
    torch, dev = batch["torch"], batch["dev"]
    ref_reads, ref_frags, ref_ctr = run(batch)
    n = batch["n"]
    p = abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_len=1000, min_q=10.0,
                        head_trim=0, tail_trim=0, max_batch_bases=batch["bases"] + 64, max_batch_reads=n,
                        max_read_len=int(batch["h_lens"].max()))
    fcap = batch["bases"] // 1000 + n + 16
    ctxs = [capi.Context(p, 0) for _ in range(3)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    outs = [(torch.empty(n * 32, dtype=torch.uint8, device=dev), torch.zeros(fcap * 24, dtype=torch.uint8, device=dev),
             torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(3)]
    torch.cuda.synchronize()
    for rep in range(2):
        for c, st, (r, f, nf) in zip(ctxs, streams, outs):
            c.submit_device(batch["seq"].data_ptr(), batch["qual"].data_ptr(), batch["offsets"].data_ptr(),
                            batch["lengths"].data_ptr(), n, batch["n_bytes"], r.data_ptr(), f.data_ptr(), fcap,
                            nf.data_ptr(), st.cuda_stream)
    for c in ctxs:
        c.wait()
    torch.cuda.synchronize()
    for c, (r, f, nf) in zip(ctxs, outs):
        reads = r.cpu().numpy().view(abi.READ_RESULT_DTYPE)
        k = int(nf[0].item())
        frags = f.cpu().numpy().view(abi.FRAGMENT_DTYPE)[:k]
        assert np.array_equal(reads, ref_reads) and np.array_equal(frags, ref_frags)
        ctr = c.counters()
        rows = slice(abi.CTR_ROWS, abi.CTR_ROWS + 4)
        exp = ref_ctr * np.uint64(2)
        exp[rows] = ref_ctr[rows]
        assert np.array_equal(ctr, exp)
        c.close()


def test_wait_and_counters_cover_a_caller_stream(batch):
    """tgsf_wait / tgsf_counters after tgsf_submit_device on a CALLER's side stream, with nothing else synchronised:
    the contract says they block until everything submitted on the context has finished (include/tgsf.h)."""
    torch, dev = batch["torch"], batch["dev"]
    ref_reads, ref_frags, ref_ctr = run(batch)
    n = batch["n"]
    p = abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_len=1000, min_q=10.0,
                        head_trim=0, tail_trim=0, max_batch_bases=batch["bases"] + 64, max_batch_reads=n,
                        max_read_len=int(batch["h_lens"].max()))
    fcap = batch["bases"] // 1000 + n + 16
    ctx = capi.Context(p, 0)
    side = torch.cuda.Stream(device=dev)
    d_reads = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_frags = torch.zeros(fcap * 24, dtype=torch.uint8, device=dev)
    d_nf = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.submit_device(batch["seq"].data_ptr(), batch["qual"].data_ptr(), batch["offsets"].data_ptr(), batch["lengths"].data_ptr(), n,
                      batch["n_bytes"], d_reads.data_ptr(), d_frags.data_ptr(), fcap, d_nf.data_ptr(), side.cuda_stream)
    ctr = ctx.counters()                    # no stream or device synchronisation by the caller
    assert np.array_equal(ctr, ref_ctr)
    ctx.wait()
    assert np.array_equal(d_reads.cpu().numpy().view(abi.READ_RESULT_DTYPE), ref_reads)
    ctx.close()


# ---- config C3's shape: HiFi reads, all of them survive the gate (the clean tables are tallied "by difference"),
# ---- blunt adapter at both ends and -- far more often than in real data, so that splits are exercised -- in the middle
N_HIFI = 262144        # 4.7 Gbases


@pytest.fixture(scope="module")
def hifi_batch():
    import torch
    import bench
    dev = torch.device("cuda", 0)
    b = bench.gen_batch(torch, dev, N_HIFI, 9, 18000.0, 2_000_000, "hifi", hifi_rates=(0.01, 0.01, 0.004))
    b["torch"], b["dev"] = torch, dev
    return b


HIFI_KW = dict(adapters=[synth.PACBIO_BLUNT, synth.PACBIO_BLUNT_RC], min_len=1000, min_q=20.0, head_trim=0, tail_trim=0,
               mid_match_len=35, extra_len=50)


def run_hifi(batch, env=None):
    torch, dev = batch["torch"], batch["dev"]
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        n = batch["n"]
        p = abi.make_params("hifi", max_batch_bases=batch["bases"] + 64, max_batch_reads=n,
                            max_read_len=int(batch["h_lens"].max()), **HIFI_KW)
        ctx = capi.Context(p, 0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    fcap = batch["bases"] // 1000 + n + 16
    d_reads = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_frags = torch.zeros(fcap * 24, dtype=torch.uint8, device=dev)
    d_nf = torch.zeros(4, dtype=torch.int32, device=dev)
    ctx.submit_device(batch["seq"].data_ptr(), batch["qual"].data_ptr(), batch["offsets"].data_ptr(), batch["lengths"].data_ptr(), n,
                      batch["n_bytes"], d_reads.data_ptr(), d_frags.data_ptr(), fcap, d_nf.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    ctx.wait()
    nf = int(d_nf[0].item())
    reads = d_reads.cpu().numpy().view(abi.READ_RESULT_DTYPE)
    frags = d_frags.cpu().numpy().view(abi.FRAGMENT_DTYPE)[:nf].copy()
    ctr = ctx.counters()
    ctx.close()
    return reads, frags, ctr


def test_hifi_full_size_invariances_and_oracle_slice(hifi_batch):
    ref = run_hifi(hifi_batch)
    reads, frags, ctr = ref
    # default strategy for this shape is "difference": the direct tally must give the same tables; so must other segmentings
    assert same(ref, run_hifi(hifi_batch, {"TGSF_CLEAN_TABLES": "direct"}))
    assert same(ref, run_hifi(hifi_batch, {"TGSF_CLEAN_TABLES": "difference"}))
    assert same(ref, run_hifi(hifi_batch, {"TGSF_SEG_COLS": "512"}))
    n, lens = hifi_batch["n"], hifi_batch["h_lens"].astype(np.int64)
    drop = ctr[abi.CTR_DROPINFO:abi.CTR_DROPINFO + 17].astype(np.int64)
    assert drop[0] + drop[2:10].sum() == n
    kept = frags["len"][(frags["flags"] & abi.FF_PASS) != 0].astype(np.int64).sum()
    assert lens.sum() == drop[1] + drop[10] + drop[12] + drop[14] + kept
    split = np.nonzero(reads["n_frags"] >= 2)[0]
    assert split.size > 200                                   # reads cut at a middle adapter
    assert drop[2] + drop[3] + drop[4] + drop[6] > 500        # classes with a middle hit
    import bench
    rng = np.random.default_rng(12)
    must = rng.choice(split, size=150, replace=False)
    n_chk, _ = bench.oracle_slice_check(hifi_batch["torch"], hifi_batch, HIFI_KW, "hifi", reads, frags, m=350, seed=6, must_include=must)
    assert n_chk >= 350


def test_bench_two_ranks_share_one_gpu_same_totals(tmp_path):
    """bench.py's N > 1 bookkeeping on the 1-GPU box: the fixed job of K batches dealt over two ranks
    (both on device 0, the tally exchange through torch.distributed/gloo -- RCCL refuses two ranks on one GPU), one
    all-reduce, must end with the SAME merged tally vector as one rank running the K batches; strong scaling is declared."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--no-e2e", "--no-cpu-baseline", "--no-oracle-check", "--reads", "4096", "--kernel-warmup", "1", "--streams", "2"]
    d1, d2 = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--kernel-steps", "6", "--detail-file", d1] + common, capture_output=True, timeout=900)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    l1 = one.stdout.decode().strip().splitlines()
    assert len(l1) == 1 and len(l1[0]) <= 4096 and json.loads(l1[0])["roofline"]["frac"] > 0      # ONE small line on stdout
    j1 = json.load(open(d1))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                          "--job-steps", "6", "--detail-file", d2] + common, capture_output=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr.decode()[-2000:]
    c2 = json.loads([l for l in two.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert c2["n_gpus"] == 2 and c2["scaling"] == "strong"
    j2 = json.load(open(d2))
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and "strong" in j2["kernel_path"]["scaling"]
    assert j2["kernel_path"]["tallies"] == j1["kernel_path"]["tallies"]
    assert j2["kernel_path"]["tallies"]["reads"] == 6 * 4096
    assert "all-reduce" in j2["kernel_path"]["tally_exchange"] and j2["kernel_path"]["rccl_ranks"] is None


def test_bench_two_ranks_end_to_end_leg_runs_one_process_per_gpu(tmp_path):
    """bench.py --gpus N: the end-to-end leg runs the command line as N rank processes (tgsfilter --ranks N, a part file
    each) -- here N = 2 sharing device 0, on a small file: the line says so, carries the per-rank SHARD lines and a value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TGSF_DEBUG_KNOBS", None)                     # (the bench runs the product as a user would)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29543", os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                          "--e2e-reads", "3000", "--steps", "2", "--warmup", "1", "--no-kernel-path", "--detail-file", str(tmp_path / "d.json")],
                         capture_output=True, timeout=1200, env=env)
    assert two.returncode == 0, two.stderr.decode()[-3000:]
    c = json.loads([l for l in two.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert c["n_gpus"] == 2 and c["steps"] == 2 and c["value"] > 0 and "cpu_baseline" not in c and "--ranks 2" in c["config"]["workload"]
    j = json.load(open(str(tmp_path / "d.json")))
    s = j["e2e"]["sinks"]["tmpfs_file"]
    assert j["n_gpus"] == 2 and j["e2e"]["ranks"] == 2 and s["ranks"] == 2 and j["steps"] == 2 and j["value"] > 0
    assert len(s["per_file"][0]["shard_lines"]) == 2 and "tgsfilter --ranks 2" in j["e2e"]["workload_long"]
    assert j["e2e"]["sinks"]["dev_null"]["same_counters_as_the_file_run"] and "cpu_baseline" not in j
