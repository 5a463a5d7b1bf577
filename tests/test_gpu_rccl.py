"""libtgsf_rccl (include/tgsf_rccl.h): the tally all-reduce of a multi-GPU job, on real RCCL.
With one GPU visible: a one-rank communicator (the sum must give the vector back, 'rows used' words included).
With two or more: two processes, one per GPU, each filters its shard; both must end with the oracle's totals."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


# (the two-GPU test first: a box with several GPUs reaches it early under -x)
WORKER = r'''
import os, sys, pickle
import numpy as np
sys.path.insert(0, os.environ["TGSF_ROOT"])
import torch
from tgsfilter_amd import abi, capi, rccl, synth, dist as tdist
from tests import parity
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
reads = synth.make_reads(77, 300, "ont", mean_len=3000, zoo=True, pmid=0.1)
p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0, head_trim=5), reads)
lo, hi = tdist.shard_range(len(reads), rank, world)
ctx = capi.Context(p, rank)
seq, qual, off, ln = synth.pack(reads[lo:hi])
res, frags = ctx.submit(seq, qual, off[:-1].copy(), ln)
idf = os.environ["TGSF_OUT"] + ".id"
if rank == 0:
    open(idf + ".tmp", "wb").write(rccl.unique_id()); os.rename(idf + ".tmp", idf)
else:
    import time
    while not os.path.exists(idf): time.sleep(0.05)
comm = rccl.comm_init_rank(open(idf, "rb").read(), rank, world)
rccl.allreduce_counters(ctx, comm, rank, world)
pickle.dump(dict(total=ctx.counters()), open(os.environ["TGSF_OUT"] + ".%d" % rank, "wb"))
rccl.comm_destroy(comm)
'''


def test_two_gpus_two_processes(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU)")
    from oracle import orc
    from tests import parity
    from tgsfilter_amd import abi, synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "out")
    env = dict(os.environ, TGSF_ROOT=ROOT, TGSF_OUT=out, WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    parts = [pickle.load(open(out + ".%d" % r, "rb")) for r in range(2)]
    reads = synth.make_reads(77, 300, "ont", mean_len=3000, zoo=True, pmid=0.1)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0, head_trim=5), reads)
    seq, qual, off, ln = synth.pack(reads)
    _, _, ectr = orc.filter_batch(p, seq, qual, off, ln, n_bins=abi.n_bins(p.max_read_len))
    assert np.array_equal(parts[0]["total"], parts[1]["total"])
    assert np.array_equal(parts[0]["total"], ectr)


def test_library_loads_and_exports():
    from tgsfilter_amd import rccl
    lib = rccl.load()
    assert hasattr(lib, "tgsf_rccl_allreduce_counters") and hasattr(lib, "tgsf_rccl_last_error")


def test_one_rank_communicator_is_identity():
    import torch
    from tests import parity
    from tgsfilter_amd import abi, capi, rccl, synth
    torch.cuda.set_device(0)
    reads = synth.make_reads(5, 200, "ont", mean_len=4000, zoo=True, pmid=0.1)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0), reads)
    ctx = capi.Context(p, 0)
    seq, qual, off, ln = synth.pack(reads)
    ctx.submit(seq, qual, off[:-1].copy(), ln)
    before = ctx.counters()
    comm = rccl.comm_init_rank(rccl.unique_id(), 0, 1)
    rccl.allreduce_counters(ctx, comm, 0, 1)
    after = ctx.counters()
    rccl.comm_destroy(comm)
    ctx.close()
    assert np.array_equal(before, after)
    assert before[abi.CTR_ROWS:abi.CTR_ROWS + 4].min() > 0
