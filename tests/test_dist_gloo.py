"""N > 1 path on CPU: two gloo ranks each filter their shard of the reads (through the serial
emulation of the kernels, since there is no GPU here) and all-reduce the tallies; the merged
tallies and the union of the per-read results must equal the oracle run over all reads."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
import numpy as np
sys.path.insert(0, os.environ["TGSF_ROOT"])
import torch.distributed as dist
from tgsfilter_amd import abi, capi, synth, dist as tdist
from tests import parity
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
reads = synth.make_reads(77, 90, "ont", mean_len=3000, zoo=True, pmid=0.1)
p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0, head_trim=5), reads)
lo, hi = tdist.shard_range(len(reads), rank, world)
ctx = capi.Context(p, 0, os.environ["TGSF_EMUL_LIB"])
seq, qual, off, ln = synth.pack(reads[lo:hi])
res, frags = ctx.submit(seq, qual, off[:-1].copy(), ln)
tdist.check_layout(ctx.ctr_words)                    # setup-time check, not part of the job's exchange
total = tdist.allreduce_counters(ctx.counters())     # ONE sum all-reduce
pickle.dump(dict(lo=lo, hi=hi, res=res, frags=frags, total=total), open(os.environ["TGSF_OUT"] + ".%d" % rank, "wb"))
dist.destroy_process_group()
'''


def test_two_rank_shard_and_allreduce(tmp_path):
    import pickle
    from oracle import orc
    from tests import parity
    from tgsfilter_amd import abi, synth

    emul_dir = os.path.join(ROOT, "tests", "emul")
    subprocess.run(["make", "-s", "-C", emul_dir], check=True)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "out")
    env = dict(os.environ, TGSF_ROOT=ROOT, TGSF_EMUL_LIB=os.path.join(emul_dir, "libtgsf_emul.so"), TGSF_OUT=out,
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=300) == 0
    parts = [pickle.load(open(out + ".%d" % r, "rb")) for r in range(2)]

    reads = synth.make_reads(77, 90, "ont", mean_len=3000, zoo=True, pmid=0.1)
    p = parity.sized(abi.make_params("ont", adapters=[synth.ONT_RAPID, synth.ONT_RAPID_RC], min_q=9.0, head_trim=5), reads)
    seq, qual, off, ln = synth.pack(reads)
    eres, efrags, ectr = orc.filter_batch(p, seq, qual, off, ln, n_bins=abi.n_bins(p.max_read_len))
    assert np.array_equal(parts[0]["total"], parts[1]["total"])
    assert np.array_equal(parts[0]["total"], ectr)
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == len(reads)
    for name in ("sum_q", "flags", "n_frags", "trimmed"):
        got = np.concatenate([pp["res"][name] for pp in parts])
        assert np.array_equal(got, eres[name]), name
    got_frags = np.concatenate([pp["frags"][["start", "len", "flags", "sum_q"]] for pp in parts])
    assert np.array_equal(got_frags, efrags[["start", "len", "flags", "sum_q"]])


MISMATCH_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["TGSF_ROOT"])
import torch.distributed as dist
from tgsfilter_amd import dist as tdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    tdist.check_layout(2000 + 100 * rank)   # layouts differ between the ranks
    code = 1
except ValueError as e:
    code = 0 if "differ in length" in str(e) else 2
dist.destroy_process_group()
sys.exit(code)
'''


def test_allreduce_refuses_mismatched_layouts(tmp_path):
    """Ranks that built their contexts with different max_read_len hold tally vectors of different lengths:
    the setup-time layout check must refuse (on every rank) instead of summing words that mean different things."""
    script = tmp_path / "worker.py"
    script.write_text(MISMATCH_WORKER)
    env = dict(os.environ, TGSF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29612", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=300) == 0
