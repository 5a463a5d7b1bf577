#!/usr/bin/env python3
"""AddressSanitizer + UBSan over the KERNEL sources: the serial emulation (tests/emul, the same tgsf_kernels.h / tgsf_core.h /
tgsf_lib.hip compiled with -DTGSF_EMUL) is built with -fsanitize=address,undefined and the emulation parity and fuzz tests run on it --
out-of-bounds reads and writes of "device" buffers and LDS arrays, signed overflow, misaligned accesses in kernel code, which
the GPU would not report (GPU sanitizers are not available on this pool).  CPU only.   tests/manual/sanitize_emul.py [pytest -k expr]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "tgsfilter_amd", "csrc", "tgsf_lib.hip")
LIB = os.path.join(ROOT, "tests", "emul", "libtgsf_emul_asan.so")
subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-DTGSF_EMUL", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                "-fno-sanitize-recover=undefined", "-Wno-unknown-pragmas", "-ffp-contract=off", "-x", "c++", SRC, "-o", LIB], check=True)
asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           TGSF_EMUL_LIB=LIB, TGSF_DEBUG_KNOBS="1")
args = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_emul_parity.py"), os.path.join(ROOT, "tests", "test_fuzz_emul.py"), "-x", "-q", "-p", "no:cacheprovider"]
if len(sys.argv) > 1:
    args += ["-k", sys.argv[1]]
rc = subprocess.run(args, env=env, cwd=ROOT).returncode
# ... and the oracle itself (oracle/tgsf_oracle.c): its pinning tests, and the emulation suites once more with the sanitized checker
ORC = os.path.join(ROOT, "oracle", "liborc_asan.so")
subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                "-o", ORC, os.path.join(ROOT, "oracle", "tgsf_oracle.c")], check=True)
env2 = dict(env, ORC_LIB=ORC)
env2.pop("TGSF_EMUL_LIB")
if len(sys.argv) <= 1:
    rc = rc or subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_pinned.py"), os.path.join(ROOT, "tests", "test_emul_parity.py"),
                               os.path.join(ROOT, "tests", "test_fuzz_emul.py"), "-x", "-q", "-p", "no:cacheprovider"], env=env2, cwd=ROOT).returncode
sys.exit(rc)
