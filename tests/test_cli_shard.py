"""One process per GPU over one input (tgsfilter --ranks N / --shard r/N, tgsfilter_amd/host/shard.h): every rank filters
its byte range of the text and writes <out>.part<r>; the pre-pass runs on rank 0 and its constants are broadcast; the
tallies are summed at the end (RCCL all-reduce with a GPU per rank, the ranks' sockets otherwise) and rank 0 writes the
one report.  The parts, concatenated in rank order, must be byte for byte the reference's -t 1 output, the INFO lines and
the report the reference's (src/TGSFilter.cpp:1808-1842 fan-out, :3208-3213 merge).

CPU: the emulation build of the command line.  -m gpu: the real binary, ranks sharing device 0 (and, at the end of the
file, a 24 000-read file against the reference binary run side by side)."""
import gzip
import json
import os
import subprocess

import pytest

from tests import cli_check, hostmodel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")
GPU_BINARY = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
SHARDABLE = [n for n in hostmodel.GOLDEN_CASES + hostmodel.GOLDEN_CLI_ONLY if cli_check.shardable(GOLD, n)]


@pytest.fixture(scope="module")
def binary():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host"), "emul"], check=True)
    return os.path.join(ROOT, "tests", "emul", "tgsfilter_emul")


def test_there_are_shardable_goldens():
    assert len(SHARDABLE) >= 15 and "ont_auto" in SHARDABLE and "hifi_auto" in SHARDABLE


@pytest.mark.parametrize("name", SHARDABLE)
@pytest.mark.parametrize("ranks", [2, 3])
def test_cli_sharded_golden(binary, golden_dir, name, ranks):
    cli_check.run_case(binary, golden_dir, name, ranks=ranks)


@pytest.mark.parametrize("name", ["ont_auto", "hifi_zoo", "ont_fasta", "ont_qc"] if "ont_qc" in SHARDABLE else ["ont_auto", "hifi_zoo", "ont_fasta"])
def test_cli_sharded_by_another_launcher(binary, golden_dir, name):
    """--shard r/N --rendezvous <path>: the ranks are started one by one (as torchrun / mpirun would) and meet at a unix socket."""
    cli_check.run_case(binary, golden_dir, name, ranks=3, launcher="external")


@pytest.mark.parametrize("name", ["down_gd", "down_r", "down_R", "down_F", "fasta_down"])
@pytest.mark.parametrize("how", ["text", "packed"])
def test_cli_sharded_downsampling(binary, golden_dir, name, how, monkeypatch):
    """A downsampling run selects among ALL reads of the job: rank 0 makes the reference's selection over every rank's kept
    fragments and hands each rank its keep flags; the second (QC) pass and the writing are per rank (the kept reads read in
    place from the text or packed; written by threads into a mapping of the part file reserved early, or by the single-stream
    writer), its tallies summed on rank 0."""
    monkeypatch.setenv("TGSF_DOWN_QC", how)
    monkeypatch.setenv("TGSF_DOWN_MAP_MIN", "1" if how == "text" else "1000000000000")
    if how == "text":
        monkeypatch.setenv("TGSF_DOWN_EARLY_MIN", "1")
        monkeypatch.setenv("TGSF_STRIDE_BYTES", "40000")
    cli_check.run_case(binary, golden_dir, name, ranks=3, extra_args=["-t", "6"])


@pytest.mark.parametrize("name", ["ont_zoo", "down_gd", "ont_fasta"])
def test_cli_sharded_gzip_output(binary, golden_dir, name, tmp_path):
    """-o <name>.gz: every rank writes gzip members into its part; the parts concatenated are one gzip stream of the
    single process's records."""
    cmd = json.load(open(os.path.join(golden_dir, name + ".cmd.json")))
    fmt = cmd.get("in_format", "fq")
    fin = tmp_path / ("in." + fmt)
    fin.write_bytes(gzip.open(os.path.join(golden_dir, "%s.in.%s.gz" % (name, fmt)), "rb").read())
    out = tmp_path / ("out.%s.gz" % ("fa" if fmt == "fa" else "fq"))
    args = [binary, "-i", str(fin), "-o", str(out)] + cmd["flags"].split()
    if cmd.get("adapters"):
        fa = tmp_path / "ad.fa"
        fa.write_text("".join(">a%d\n%s\n" % (i, a) for i, a in enumerate(cmd["adapters"])))
        args += ["-a", str(fa)]
    p = subprocess.run(args + ["--ranks", "3"], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    cat = b"".join((tmp_path / ("%s.part%d" % (out.name, r))).read_bytes() for r in range(3))
    assert gzip.decompress(cat) == gzip.open(os.path.join(golden_dir, name + ".out.fq.gz"), "rb").read()


def test_cli_sharded_ranks_hand_their_stderr_lines_over_whole(binary, golden_dir, tmp_path):
    """The ranks of a job share one stderr: every line -- rank 0's INFO lines, every rank's TGSF_TIMING lines -- is one write,
    none lands in the middle of another (bench.py reads the INFO, SHARD and TIMING lines of a job)."""
    fin = tmp_path / "in.fq"
    fin.write_bytes(gzip.open(os.path.join(golden_dir, "ont_zoo.in.fq.gz"), "rb").read())
    known = ("INFO:", "SHARD ", "TIMING:", "POOL:", "GPU:", "RESERVE:", "DEVICE ", "DOWN:", "CLOCK:", "Warning:", "PREPASS:", "CPU:")
    for _ in range(25):
        p = subprocess.run([binary, "-i", str(fin), "-o", str(tmp_path / "o.fq"), "-x", "ont", "-t", "12", "--ranks", "6"],
                           capture_output=True, timeout=120, env=dict(os.environ, TGSF_TIMING="1"))
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        lines = p.stderr.decode().splitlines()
        assert all(l.startswith(known) for l in lines), [l for l in lines if not l.startswith(known)][:3]
        assert sum(l.startswith("SHARD ") for l in lines) == 6


def test_cli_eight_ranks_on_eight_devices_bind_to_their_numa_nodes(binary, golden_dir, tmp_path):
    """VERDICT r5 item 7, a dry run of the first 8-GPU sitting on the emulation: `--ranks 8 --devices 0,...,7` with a made-up
    place for every device (bus id emul:<n>; devices 0-3 on NUMA node 0, 4-7 on node 1) and a stubbed sysfs.  The ranks'
    bus ids are gathered, rank 0 finds a GPU per rank, every rank binds its feeders to the CPUs of its device's node
    (tgsf_device_location -> node<N>/cpulist -> sched_setaffinity, within what the process may use), and the eight parts,
    concatenated, are the single process's bytes."""
    raw = gzip.open(os.path.join(golden_dir, "ont_zoo.in.fq.gz"), "rb").read()
    fin = tmp_path / "in.fq"
    fin.write_bytes(raw)
    nodes = tmp_path / "nodes"
    ncpu = len(os.sched_getaffinity(0))
    cpus = sorted(os.sched_getaffinity(0))
    half = max(1, ncpu // 2)
    for n, cs in ((0, cpus[:half]), (1, cpus[half:] or cpus[:half])):
        (nodes / ("node%d" % n)).mkdir(parents=True)
        (nodes / ("node%d" % n) / "cpulist").write_text(",".join(str(c) for c in cs) + "\n")
    common = ["-i", str(fin), "-x", "ont", "-l", "500", "-q", "7", "-t", "16"]
    env = dict(os.environ, TGSF_DEBUG_KNOBS="1", TGSF_EMUL_NUMA="0,0,0,0,1,1,1,1", TGSF_SYSFS_NODES=str(nodes), TGSF_TIMING="1")
    p1 = subprocess.run([binary, "-o", str(tmp_path / "one.fq")] + common, capture_output=True, timeout=300)
    p8 = subprocess.run([binary, "-o", str(tmp_path / "eight.fq"), "--ranks", "8", "--devices", "0,1,2,3,4,5,6,7"] + common,
                        capture_output=True, timeout=600, env=env)
    assert p1.returncode == 0 and p8.returncode == 0, p8.stderr.decode()[-2000:]
    parts = b"".join((tmp_path / ("eight.fq.part%d" % r)).read_bytes() for r in range(8))
    assert parts == (tmp_path / "one.fq").read_bytes() and len(parts) > 0
    info = lambda e: [l for l in e.decode().splitlines() if l.startswith("INFO:") and "written to" not in l]
    assert info(p1.stderr) == info(p8.stderr)
    err = p8.stderr.decode()
    for d in range(8):                                                # every rank's DEVICE line: its device, bound to that device's node
        line = [l for l in err.splitlines() if l.startswith("DEVICE %d:" % d)]
        assert len(line) == 1 and "(feeders bound to NUMA node %d)" % (0 if d < 4 else 1) in line[0], line
    assert err.count("SHARD ") == 8
    # ranks that SHARE a device are not bound (the bus-id gather says so, not the command line)
    p2 = subprocess.run([binary, "-o", str(tmp_path / "two.fq"), "--ranks", "2", "--devices", "3"] + common, capture_output=True, timeout=600, env=env)
    assert p2.returncode == 0 and "feeders bound" not in p2.stderr.decode()


def test_cli_one_rank_is_a_job_too(binary, golden_dir):
    cli_check.run_case(binary, golden_dir, "ont_zoo", ranks=1)


def test_cli_sharded_more_ranks_than_reads(binary, golden_dir, tmp_path):
    """Ranks whose byte range holds no record start write an empty part; the job's output is still the single process's."""
    raw = gzip.open(os.path.join(golden_dir, "ont_zoo.in.fq.gz"), "rb").read()
    recs = raw.split(b"\n")
    fin = tmp_path / "in.fq"
    fin.write_bytes(b"\n".join(recs[:4 * 5]) + b"\n")                # five reads
    common = ["-i", str(fin), "-x", "ont", "-l", "500", "-q", "7", "-5", "0", "-3", "0"]
    p1 = subprocess.run([binary, "-o", str(tmp_path / "one.fq")] + common, capture_output=True, timeout=300)
    # an earlier, larger job of this tool (11 ranks) leaves its parts and its list; a file of the user's that merely has such a name
    p11 = subprocess.run([binary, "-o", str(tmp_path / "nine.fq"), "--ranks", "11"] + common, capture_output=True, timeout=300)
    assert p11.returncode == 0 and len(list(tmp_path.glob("nine.fq.part*"))) == 12             # 11 parts + the list
    assert (tmp_path / "nine.fq.parts").read_text().splitlines()[1:] == [str(tmp_path / ("nine.fq.part%d" % k)) for k in range(11)]
    (tmp_path / "nine.fq.part12").write_bytes(b"@mine\nA\n+\n!\n")
    p9 = subprocess.run([binary, "-o", str(tmp_path / "nine.fq"), "--ranks", "9"] + common, capture_output=True, timeout=300)
    assert p1.returncode == 0 and p9.returncode == 0, p9.stderr.decode()[-2000:]
    # parts 9 and 10 -- named by the earlier job's list -- are gone; part12, which no list names, stays, with a warning
    assert sorted(f.name for f in tmp_path.glob("nine.fq.part*")) == sorted(["nine.fq.part%d" % k for k in list(range(9)) + [12]] + ["nine.fq.parts"])
    assert p9.stderr.count(b"of an earlier job with more ranks, was removed") == 2
    assert p9.stderr.count(b"nine.fq.part12 exists and is not a part of this job") == 1
    assert (tmp_path / "nine.fq.parts").read_text().splitlines()[1:] == [str(tmp_path / ("nine.fq.part%d" % k)) for k in range(9)]
    parts = b"".join((tmp_path / ("nine.fq.part%d" % r)).read_bytes() for r in range(9))
    assert parts == (tmp_path / "one.fq").read_bytes() and len(parts) > 0
    info = lambda e: [l for l in e.decode().splitlines() if l.startswith("INFO:") and "written to" not in l]
    assert info(p1.stderr) == info(p9.stderr)


def _adversarial_fastq(n=400, seed=3, crlf=False):
    """Quality lines that begin with '@' and '+', headers that contain '+' and '@': every line start looks like a record start."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(rng.integers(600, 2500))
        seq = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)])
        q = bytearray(rng.integers(40, 75, L).astype(np.uint8).tobytes())
        q[0] = ord("@") if i % 2 == 0 else ord("+")
        if L > 1:
            q[1] = ord("@")
        out.append(b"@read%d +@ x\n" % i + seq + b"\n+\n" + bytes(q) + b"\n")
    text = b"".join(out)
    return text.replace(b"\n", b"\r\n") if crlf else text


@pytest.mark.parametrize("crlf", [False, True])
@pytest.mark.parametrize("ranks", [2, 5, 16])
def test_cli_sharded_cuts_on_adversarial_text(binary, tmp_path, ranks, crlf):
    fin = tmp_path / "in.fq"
    fin.write_bytes(_adversarial_fastq(crlf=crlf))
    common = ["-i", str(fin), "-x", "ont", "-l", "500", "-q", "7", "-5", "3", "-3", "2", "-t", "4"]
    p1 = subprocess.run([binary, "-o", str(tmp_path / "one.fq")] + common, capture_output=True, timeout=600)
    pn = subprocess.run([binary, "-o", str(tmp_path / "n.fq"), "--ranks", str(ranks)] + common, capture_output=True, timeout=600)
    assert p1.returncode == 0 and pn.returncode == 0, pn.stderr.decode()[-2000:]
    parts = b"".join((tmp_path / ("n.fq.part%d" % r)).read_bytes() for r in range(ranks))
    assert parts == (tmp_path / "one.fq").read_bytes() and parts.count(b"\n") >= 4 * 300


def test_cli_sharded_refuses_a_text_it_cannot_cut_like_the_sequential_reader(binary, tmp_path):
    """Stray lines between two records are skipped by the reference's reader within its five attempts per record
    (src/TGSFilter.cpp:689-698): a reader that starts behind them has not seen them.  Where a cut falls next to such lines
    the job stops with a message instead of reading the text its own way; a malformed record inside a part stops it as well."""
    text = _adversarial_fastq(n=60)
    recs = text.split(b"@read")
    mid = len(recs) // 2
    broken = b"@read".join(recs[:mid]) + b"stray line\n@read" + b"@read".join(recs[mid:])
    fin = tmp_path / "in.fq"
    fin.write_bytes(broken)
    common = ["-i", str(fin), "-x", "ont", "-l", "500", "-q", "7", "-5", "0", "-3", "0"]
    p1 = subprocess.run([binary, "-o", str(tmp_path / "one.fq")] + common, capture_output=True, timeout=300)
    assert p1.returncode == 0                                        # (the sequential reader skips the stray line)
    bad = 0
    for ranks in (2, 3, 4, 6):
        pn = subprocess.run([binary, "-o", str(tmp_path / ("n%d.fq" % ranks)), "--ranks", str(ranks)] + common, capture_output=True, timeout=300)
        if pn.returncode != 0:
            bad += 1
            assert b"cannot be cut near byte" in pn.stderr
        else:                                                        # the stray line fell inside a part: read as the sequential reader reads it
            parts = b"".join((tmp_path / ("n%d.fq.part%d" % (ranks, r))).read_bytes() for r in range(ranks))
            assert parts == (tmp_path / "one.fq").read_bytes()
    # a record whose quality line is short: the sequential reader ends the stream there (:719-723); a part that holds it
    # and is not the last cannot end where the next begins
    lines = text.split(b"\n")
    lines[4 * 10 + 3] = lines[4 * 10 + 3][:-5]
    fin.write_bytes(b"\n".join(lines))
    pn = subprocess.run([binary, "-o", str(tmp_path / "m.fq"), "--ranks", "3"] + common, capture_output=True, timeout=300)
    assert pn.returncode != 0 and b"cannot be cut near byte" in pn.stderr


def test_cli_sharded_refusals(binary, golden_dir, tmp_path):
    raw = gzip.open(os.path.join(golden_dir, "ont_zoo.in.fq.gz"), "rb").read()
    fin = tmp_path / "in.fq"
    fin.write_bytes(raw)
    gz = tmp_path / "in2.fq.gz"
    gz.write_bytes(gzip.compress(raw))
    base = [binary, "-x", "ont", "-o", str(tmp_path / "o.fq")]
    for extra, what in ((["-i", str(gz), "--ranks", "2"], b"plain FASTQ"),
                        (["-i", str(fin), "--shard", "0/2"], b"--rendezvous"),
                        (["-i", str(fin), "--shard", "2/2", "--rendezvous", str(tmp_path / "s")], b"no such rank"),
                        (["-i", str(fin), "--ranks", "2", "--shard", "0/2", "--rendezvous", str(tmp_path / "s")], b"--ranks starts the ranks itself")):
        p = subprocess.run(base + extra, capture_output=True, timeout=120)
        assert p.returncode != 0 and what in p.stderr, (extra, p.stderr)
    p = subprocess.run([binary, "-x", "ont", "-i", str(fin), "--ranks", "2"], capture_output=True, timeout=120)
    assert p.returncode != 0 and b"-o is needed" in p.stderr
    # a rendezvous path that names a file of the user's is refused, and the file is still there
    p = subprocess.run(base + ["-i", str(fin), "--shard", "0/2", "--rendezvous", str(gz)], capture_output=True, timeout=120)
    assert p.returncode != 0 and b"is not a socket" in p.stderr and gz.exists()


def test_cli_shard_env_as_torchrun_sets_it(binary, golden_dir, tmp_path):
    """--shard env: RANK / WORLD_SIZE (and LOCAL_RANK for the device) from the environment."""
    raw = gzip.open(os.path.join(golden_dir, "hifi_auto.in.fq.gz"), "rb").read()
    fin = tmp_path / "in.fq"
    fin.write_bytes(raw)
    cmd = json.load(open(os.path.join(golden_dir, "hifi_auto.cmd.json")))
    assert not cmd["adapters"]
    # (no --rendezvous: the ranks meet at a socket named after the launcher's MASTER_PORT)
    args = [binary, "-i", str(fin), "-o", str(tmp_path / "o.fq"), "-t", "1"] + cmd["flags"].split() + ["--shard", "env"]
    port = str(20000 + os.getpid() % 20000)
    procs = [subprocess.Popen(args, stderr=subprocess.PIPE, env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_PORT=port)) for r in range(2)]
    errs = [q.communicate(timeout=600)[1] for q in procs]
    assert all(q.returncode == 0 for q in procs), errs
    parts = b"".join((tmp_path / ("o.fq.part%d" % r)).read_bytes() for r in range(2))
    assert parts == gzip.open(os.path.join(golden_dir, "hifi_auto.out.fq.gz"), "rb").read()
    assert b"INFO:" in errs[0] and b"INFO:" not in errs[1]           # rank 0 speaks for the job
    # the default rendezvous is in a directory of this user's alone -- $XDG_RUNTIME_DIR when it is one, else
    # /tmp/tgsfilter-<uid> (0700) --, never a predictable name straight in /tmp
    assert not os.path.exists("/tmp/tgsfilter.%s.sock" % port)
    xdg = os.environ.get("XDG_RUNTIME_DIR")
    if not (xdg and os.path.isdir(xdg) and os.stat(xdg).st_uid == os.geteuid() and os.stat(xdg).st_mode & 0o77 == 0):
        d = "/tmp/tgsfilter-%d" % os.geteuid()
        assert os.path.isdir(d) and os.stat(d).st_mode & 0o777 == 0o700


def test_cli_shard_a_live_rendezvous_socket_is_not_taken_away(binary, golden_dir, tmp_path):
    """rank 0 removes a socket a killed job left behind -- but not one somebody is still listening on (another job)."""
    import socket
    fin = tmp_path / "in.fq"
    fin.write_bytes(gzip.open(os.path.join(golden_dir, "ont_zoo.in.fq.gz"), "rb").read())
    path = str(tmp_path / "rdv.sock")
    live = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    live.bind(path)
    live.listen(4)
    args = [binary, "-i", str(fin), "-o", str(tmp_path / "o.fq"), "-x", "ont", "--shard", "0/2", "--rendezvous", path]
    p = subprocess.run(args, capture_output=True, timeout=120)
    assert p.returncode != 0 and b"another job is listening" in p.stderr and os.path.exists(path)
    live.close()                                                      # now it is a stale one: it goes, and the job waits for its ranks
    q = subprocess.Popen(args, stderr=subprocess.PIPE)
    r1 = subprocess.run(args[:-4] + ["--shard", "1/2", "--rendezvous", path], capture_output=True, timeout=300)
    err0 = q.communicate(timeout=300)[1]
    assert q.returncode == 0 and r1.returncode == 0, (err0[-1000:], r1.stderr[-1000:])


def test_cli_sharded_job_ends_when_a_rank_fails(binary, golden_dir, tmp_path):
    """A rank that ends with an error takes the job with it (no rank waits for ever for a peer that is gone)."""
    raw = gzip.open(os.path.join(golden_dir, "ont_zoo.in.fq.gz"), "rb").read()
    fin = tmp_path / "in.fq"
    fin.write_bytes(raw)
    # -q above the sample's qualities ends rank 0 in the pre-pass (Get_qType :1060-1065)
    p = subprocess.run([binary, "-i", str(fin), "-o", str(tmp_path / "o.fq"), "-x", "ont", "-q", "90", "--ranks", "3"], capture_output=True, timeout=120)
    assert p.returncode != 0 and b"Please reset -q parameter" in p.stderr


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", SHARDABLE)
def test_cli_gpu_sharded_golden(golden_dir, name):
    """Three rank processes sharing device 0 (tallies summed over the ranks' sockets: RCCL refuses two ranks on one device)."""
    cli_check.run_case(GPU_BINARY, golden_dir, name, ranks=3, extra_args=["--devices", "0"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ont_auto", "hifi_auto", "ont_fasta"])
def test_cli_gpu_one_rank_all_reduces_over_rccl(golden_dir, name, monkeypatch):
    """--ranks 1 with a GPU of its own: the communicator is set up beside the filtering and the tallies go through the RCCL
    all-reduce (libtgsf_rccl) -- a one-rank communicator on this box; the program path of a GPU per rank."""
    if not os.path.exists(os.path.join(ROOT, "tgsfilter_amd", "libtgsf_rccl.so")):
        pytest.skip("libtgsf_rccl.so not built (no librccl)")
    monkeypatch.setenv("TGSF_SHARD_EXCHANGE", "rccl")                  # (no silent fall-back to the sockets)
    cli_check.run_case(GPU_BINARY, golden_dir, name, ranks=1)


@pytest.mark.gpu
def test_cli_gpu_a_communicator_that_does_not_come_up_does_not_hold_the_job(golden_dir, tmp_path):
    """The RCCL communicator is set up by a helper beside the filtering.  One that is still not there some time after the
    filtering is over (here: a helper that sleeps, and no patience) is given up: a warning, the tallies summed over the
    ranks' sockets, the same output -- and the run does not wait for the helper."""
    import time
    if not os.path.exists(os.path.join(ROOT, "tgsfilter_amd", "libtgsf_rccl.so")):
        pytest.skip("libtgsf_rccl.so not built (no librccl)")
    raw = gzip.open(os.path.join(golden_dir, "hifi_auto.in.fq.gz"), "rb").read()
    fin = tmp_path / "in.fq"
    fin.write_bytes(raw)
    cmd = json.load(open(os.path.join(golden_dir, "hifi_auto.cmd.json")))
    env = dict(os.environ, TGSF_DEBUG_KNOBS="1", TGSF_RCCL_STALL_S="60", TGSF_RCCL_INIT_TIMEOUT_S="0.5")
    env.pop("TGSF_SHARD_EXCHANGE", None)
    t0 = time.time()
    p = subprocess.run([GPU_BINARY, "-i", str(fin), "-o", str(tmp_path / "o.fq"), "-t", "1"] + cmd["flags"].split() + ["--ranks", "1"],
                       capture_output=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert time.time() - t0 < 45
    assert b"the communicator was not up" in p.stderr and b"summed over the ranks' sockets" in p.stderr
    assert (tmp_path / "o.fq.part0").read_bytes() == gzip.open(os.path.join(golden_dir, "hifi_auto.out.fq.gz"), "rb").read()
    ref_info = [l for l in open(os.path.join(golden_dir, "hifi_auto.stderr.txt")).read().splitlines() if l.startswith("INFO:") and "written to" not in l]
    assert [l for l in p.stderr.decode().splitlines() if l.startswith("INFO:") and "written to" not in l] == ref_info


@pytest.mark.gpu
def test_cli_gpu_an_all_reduce_that_hangs_does_not_hold_the_job_and_rccl_is_checked(golden_dir, tmp_path):
    """the tally all-reduce runs on a helper thread under a deadline of its own (here: a helper that sleeps, and
    no patience): a warning, the sum over the ranks' sockets, the same output.  And when the collective does run, rank 0
    compares its result with that sum and says so in the SHARD line."""
    import time
    if not os.path.exists(os.path.join(ROOT, "tgsfilter_amd", "libtgsf_rccl.so")):
        pytest.skip("libtgsf_rccl.so not built (no librccl)")
    fin = tmp_path / "in.fq"
    fin.write_bytes(gzip.open(os.path.join(golden_dir, "hifi_auto.in.fq.gz"), "rb").read())
    cmd = json.load(open(os.path.join(golden_dir, "hifi_auto.cmd.json")))
    want = gzip.open(os.path.join(golden_dir, "hifi_auto.out.fq.gz"), "rb").read()
    args = [GPU_BINARY, "-i", str(fin), "-o", str(tmp_path / "o.fq"), "-t", "1"] + cmd["flags"].split() + ["--ranks", "1"]
    env = dict(os.environ, TGSF_DEBUG_KNOBS="1", TGSF_RCCL_ALLREDUCE_STALL_S="60", TGSF_RCCL_ALLREDUCE_TIMEOUT_S="0.5", TGSF_TIMING="1")
    env.pop("TGSF_SHARD_EXCHANGE", None)
    t0 = time.time()
    p = subprocess.run(args, capture_output=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert time.time() - t0 < 45
    assert b"the tally all-reduce had not returned" in p.stderr and b"summed on rank 0 over the ranks' sockets" in p.stderr
    assert (tmp_path / "o.fq.part0").read_bytes() == want
    # ... and with TGSF_SHARD_EXCHANGE=rccl the same hang is an error, not a fall-back
    p = subprocess.run(args, capture_output=True, timeout=300, env=dict(env, TGSF_SHARD_EXCHANGE="rccl"))
    assert p.returncode != 0 and b"the tally all-reduce had not returned" in p.stderr
    # the collective as it runs: checked against the sockets' sum
    env2 = dict(os.environ, TGSF_TIMING="1", TGSF_SHARD_EXCHANGE="rccl")
    p = subprocess.run(args, capture_output=True, timeout=300, env=env2)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"RCCL all-reduce on the devices, equal to the sum over the ranks' sockets" in p.stderr
    assert (tmp_path / "o.fq.part0").read_bytes() == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ont_auto", "hifi_auto"])
def test_cli_gpu_a_rank_per_gpu_all_reduces_over_rccl(golden_dir, name, monkeypatch):
    """Needs two GPUs or more (skipped on the pool's 1-GPU boxes): `--ranks N`, a GPU per rank -- the communicator across real
    devices, the NUMA binding, a part file per rank, ONE all-reduce of the tallies over RCCL / xGMI, checked on rank 0
    against the sum over the sockets (TGSF_SHARD_EXCHANGE=rccl: any fall-back is an error)."""
    import torch
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("needs two GPUs (one process per GPU)")
    if not os.path.exists(os.path.join(ROOT, "tgsfilter_amd", "libtgsf_rccl.so")):
        pytest.skip("libtgsf_rccl.so not built (no librccl)")
    monkeypatch.setenv("TGSF_SHARD_EXCHANGE", "rccl")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cli_check.run_case(GPU_BINARY, golden_dir, name, ranks=n)


@pytest.mark.gpu
def test_cli_gpu_sharded_by_another_launcher(golden_dir):
    cli_check.run_case(GPU_BINARY, golden_dir, "ont_auto", ranks=2, launcher="external", extra_args=["--device", "0"])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")
def test_cli_gpu_three_ranks_on_a_real_file_against_the_reference(tmp_path):
    """Config C4's program path on the one device of this box: three rank processes (three contexts each) over a
    24 000-read file (C2's shape at 6 kb: 0.14 Gbases, ~290 MB of text) with the automatic pre-pass on rank 0.  The parts
    concatenated in rank order, the INFO lines and the report's tables and plotted data must equal the reference binary's
    (-t 1: input order) and the single process's."""
    import hashlib
    import tempfile
    from tgsfilter_amd import synth
    shm = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=shm) as td:
        fq = os.path.join(td, "in.fq")
        bases, _ = synth.write_ont_fastq(fq, 24_000, seed=21, mean_len=6000.0, max_len=200_000, reads_per_job=512)
        assert bases > 1e8
        common = ["-i", fq, "-x", "ont", "-l", "1000", "-q", "10"]               # automatic pre-pass, as C2
        env = dict(os.environ, TGSF_BATCH_BYTES="1500000", TGSF_EARLY_OPEN_MIN="1", TGSF_STRIDE_BYTES="8000000", TGSF_TIMING="1")

        def run(tag, exe, extra, e=None, parts=0):
            out = os.path.join(td, tag + ".fq")
            p = subprocess.run([exe, "-o", out] + common + extra, capture_output=True, env=e)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            info = [l for l in p.stderr.decode().splitlines() if l.startswith("INFO: ") and "written to" not in l]
            html = open(os.path.join(td, tag + ".html"), "rb").read()
            body = b"\n".join(l for l in html.splitlines() if b"<tr>" in l or l.lstrip().startswith(b"var data"))
            h = hashlib.sha256()
            for f in ([out] if not parts else ["%s.part%d" % (out, r) for r in range(parts)]):
                h.update(open(f, "rb").read())
            return h.hexdigest(), info, hashlib.sha256(body).hexdigest(), p.stderr.decode()

        one = run("one", GPU_BINARY, ["-t", "16", "--devices", "0"], env)
        three = run("three", GPU_BINARY, ["-t", "16", "--ranks", "3", "--devices", "0"], env, parts=3)
        ref = run("ref", REF, ["-t", "1"])
        assert three[1] == one[1] == ref[1]
        assert three[0] == one[0] == ref[0]
        assert three[2] == one[2] == ref[2]
        assert three[3].count("SHARD ") == 3 and "summed on rank 0 over the ranks' sockets" in three[3]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/tgsfilter_ref not built (make -C oracle ref)")
def test_cli_gpu_three_ranks_config5_shape_against_the_reference(tmp_path):
    """Config C5's flags (repeat gate on the GPU, longest-first downsampling) as a job of three ranks: the selection is made
    on rank 0 over every rank's kept fragments; parts concatenated, INFO lines equal the reference's (-t 1)."""
    import tempfile
    from tgsfilter_amd import synth
    shm = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=shm) as td:
        fq = os.path.join(td, "c5.fq")
        bases, _ = synth.write_ont_fastq(fq, 800, seed=5, mean_len=150000.0, max_len=2_000_000, reads_per_job=64)
        assert bases > 4e7
        fa = os.path.join(td, "rapid.fa")
        open(fa, "wb").write(b">rapid\n" + synth.ONT_RAPID + b"\n")
        common = ["-i", fq, "-x", "ont", "-l", "1000", "-q", "10", "-5", "0", "-3", "0", "-a", fa, "-p", "100", "-k", "11", "-g", "30m", "-d", "2"]
        outs = {}
        for tag, exe, extra, parts in (("ref", REF, ["-t", "1"], 0), ("ours", GPU_BINARY, ["-t", "16", "--ranks", "3", "--devices", "0"], 3)):
            out = os.path.join(td, tag + ".fq")
            p = subprocess.run([exe, "-o", out] + extra + common, capture_output=True)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            info = [l for l in p.stderr.decode().splitlines() if l.startswith("INFO: ") and "written to" not in l and "input adapter" not in l]
            data = b"".join(open(f, "rb").read() for f in ([out] if not parts else ["%s.part%d" % (out, r) for r in range(parts)]))
            outs[tag] = (data, info)
        assert outs["ours"][1] == outs["ref"][1]
        assert outs["ours"][0] == outs["ref"][0] and len(outs["ref"][0]) > 1e6
