"""Schedule fuzzing inside the GPU suite: the verification build of the library (`make fuzz`, -DXVEC_SCHED_FUZZ: every wave of
the GEMM kernels sleeps a pseudo-random time at every phase boundary, kernels.hip `sched_fuzz`) must produce the product
build's bits.  The full matrix is tools/fuzz_schedule.py (profiles/r05_schedule_fuzz.md); this is its one-minute subset."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

FUZZ_LIB = os.path.join(H.ROOT, H.PKG_NAME, "fuzz", "libxvec_hip.so")


def _need_fuzz_build():
    if not os.path.exists(FUZZ_LIB):
        pytest.skip("no fuzz build (make -C %s/csrc fuzz; __graft_entry__.build() makes it)" % H.PKG_NAME)
    # both libraries name the kernel sources they were built from (xv_version): a fuzz build of OTHER sources would compare two
    # different kernels
    import ctypes
    vers = []
    for path in (H.pkg().LIB_PATH, FUZZ_LIB):
        L = ctypes.CDLL(path)
        L.xv_version.restype = ctypes.c_char_p
        vers.append(L.xv_version().decode())
    assert "schedule-fuzzing build" in vers[1] and "schedule-fuzzing build" not in vers[0], vers
    sha = [v.split("kernels ")[1][:16] for v in vers]
    assert sha[0] == sha[1], "the fuzz library was built from other kernel sources than the product library: %s (make fuzz)" % vers


def test_forward_passes_of_the_fuzz_build_are_bit_identical():
    """v2 network, fp16mx / fp16mx2 / a mixture with lite layers, on the 256 x 256 kernel and on the 512 x 128 stream-K kernel,
    a ragged batch: three forward passes each on the fuzz build against one on the product build, compared as bits."""
    _need_fuzz_build()
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "tools", "fuzz_schedule.py"), "--smoke", "--repeats", "3", "--no-kernel-tests"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert last["cases"] == 6 and last["fuzzed_forward_passes"] == 18 and last["passes_that_differ_from_the_product_build"] == 0, last


def test_kernel_launch_tests_pass_on_the_fuzz_build():
    """tests/test_gpu_kernels.py - single launches against the bit-exact emulations - on the fuzz build: the 47 launches of the
    256 x 256 kernel (every shape of the 1.5-pass arithmetic among them, whose round-4 race this method finds in 12 of them when
    it is put back: make fuzz-inject).  All 142 launch tests on the fuzz build: tools/fuzz_schedule.py (48 s; they ran inside the
    suite in round 5, which the driver's time does not leave room for)."""
    _need_fuzz_build()
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(H.ROOT, "tests", "test_gpu_kernels.py"), "-m", "gpu", "-q", "-x",
                        "-k", "p8 and not bit_stable", "-p", "no:cacheprovider"], env=dict(os.environ, XVEC_LIB=FUZZ_LIB), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
