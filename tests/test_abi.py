"""The C-ABI library loads on a GPU-less box, exports every symbol include/tgsf.h declares, and
refuses to run without a device (there is no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from tgsfilter_amd import abi, capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "csrc")], check=True)
    return capi.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "tgsf.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(tgsf_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s)
    assert lib.tgsf_abi_version() == abi.ABI_VERSION


def test_struct_layout_matches_header():
    """sizeof() of the ctypes mirrors == what the C compiler sees."""
    src = r'''
    #include <stdio.h>
    #include "tgsf.h"
    int main(void){ printf("%zu %zu %zu %zu %zu\n", sizeof(tgsf_params), sizeof(tgsf_batch_in),
        sizeof(tgsf_batch_out), sizeof(tgsf_read_result), sizeof(tgsf_fragment)); return 0; }'''
    exe = "/tmp/tgsf_sizes"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    out = subprocess.run([exe], capture_output=True, check=True).stdout.split()
    got = [int(x) for x in out]
    assert got == [C.sizeof(abi.Params), C.sizeof(abi.BatchIn), C.sizeof(abi.BatchOut),
                   abi.READ_RESULT_DTYPE.itemsize, abi.FRAGMENT_DTYPE.itemsize]


def test_constants_match_header():
    """The constants of the tgsf_submit_device contract as the Python binding states them."""
    import re
    from tgsfilter_amd import abi
    hdr = open(os.path.join(ROOT, "include", "tgsf.h")).read()
    assert int(re.search(r"#define TGSF_MAX_ENQUEUED (\d+)", hdr).group(1)) == abi.MAX_ENQUEUED
    assert int(re.search(r"#define TGSF_NFRAGS_NOT_FINAL (0x[0-9A-Fa-f]+)u", hdr).group(1), 16) == abi.NFRAGS_NOT_FINAL
    assert int(re.search(r"#define TGSF_N_STAGES (\d+)", hdr).group(1)) == abi.N_STAGES


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = abi.make_params("ont", adapters=[synth.ONT_RAPID], max_batch_bases=1000, max_batch_reads=4, max_read_len=500)
    with pytest.raises(capi.TgsfError) as ei:
        capi.Context(p, 0)
    assert ei.value.code == abi.E_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_parameter_validation(lib):
    p = abi.make_params("ont", adapters=[b"A" * 8193], max_batch_bases=1000, max_batch_reads=4, max_read_len=500)      # beyond TGSF_MAX_ADAPTER_LEN
    with pytest.raises(capi.TgsfError) as ei:
        capi.Context(p, 0)
    assert ei.value.code == abi.E_UNSUPPORTED


def test_rccl_library_exports():
    """libtgsf_rccl.so (include/tgsf_rccl.h, the optional tally all-reduce) loads and exports what its header declares."""
    hdr = open(os.path.join(ROOT, "include", "tgsf_rccl.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(tgsf_rccl_\w+)\s*\(", hdr, flags=re.M))
    assert declared == {"tgsf_rccl_allreduce_counters", "tgsf_rccl_last_error", "tgsf_rccl_unique_id", "tgsf_rccl_comm_init",
                        "tgsf_rccl_comm_count", "tgsf_rccl_comm_destroy"}
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "csrc")], check=True)
    from tgsfilter_amd import rccl
    lib = rccl.load()
    for s in declared:
        assert hasattr(lib, s)


def test_backend_symbol_names_what_runs_the_kernels(lib):
    """include/tgsf.h: tgsf_backend() -- "hip:gfx950" for the product library, "emulation" for the tests' serial build."""
    assert lib.tgsf_backend() == b"hip:gfx950"
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul")], check=True)
    emul = capi.load(os.path.join(ROOT, "tests", "emul", "libtgsf_emul.so"))          # (an explicit path: the tests' way in)
    assert emul.tgsf_backend() == b"emulation"


def test_a_stray_TGSF_LIB_never_turns_a_gpu_run_into_a_cpu_run(tmp_path):
    """TGSF_LIB says where libtgsf.so is installed; pointing it at the emulation must not make the product
    -- the command line or the Python binding -- run on the CPU, with or without the debug switch."""
    import sys
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tgsfilter_amd", "host")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul")], check=True)
    emul = os.path.join(ROOT, "tests", "emul", "libtgsf_emul.so")
    fq = tmp_path / "in.fq"
    reads = synth.make_reads(5, 6, "ont", mean_len=3000)
    fq.write_bytes(b"".join(b"@%s\n%s\n+\n%s\n" % (n, s, q) for n, s, q in reads))
    cli = os.path.join(ROOT, "tgsfilter_amd", "bin", "tgsfilter")
    for knobs in ({}, {"TGSF_DEBUG_KNOBS": "1"}):
        env = dict(os.environ, TGSF_LIB=emul, **knobs)
        if not knobs:
            env.pop("TGSF_DEBUG_KNOBS", None)
        out = tmp_path / "out.fq"
        p = subprocess.run([cli, "-i", str(fq), "-o", str(out), "-x", "ont", "-5", "0", "-3", "0"], capture_output=True, env=env, timeout=120)
        assert p.returncode != 0 and b"emulation" in p.stderr and b"no CPU fallback" in p.stderr, p.stderr[-500:]
        assert not out.exists() or out.stat().st_size == 0
        code = "from tgsfilter_amd import capi\ntry:\n    capi.load()\nexcept RuntimeError as e:\n    assert 'not the HIP build' in str(e), e\n    raise SystemExit(7)\n"
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, env=env, cwd=ROOT, timeout=120)
        assert p.returncode == 7, p.stderr[-500:]
