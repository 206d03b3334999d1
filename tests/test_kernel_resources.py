"""Register budgets of the gfx950 kernels as hipcc reports them (cross-compiled here, no GPU): properties the timings rest
on and that change silently with the compiler's choices - a planes epilogue that spills into scratch, or a
single-pass per-tile kernel at 130 registers instead of 128, which halves the workgroups a CU holds (tdnn5 in fp16:
0.19 -> 0.28 ms, found in round 3 only because the bench line lists the opt-in modes)."""
import os
import re
import shutil
import subprocess

import pytest

import helpers as H

TOOL = os.path.join(H.ROOT, "tools", "kernel_resources.sh")


@pytest.fixture(scope="module")
def resources():
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", TOOL], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    out = {}
    for line in r.stdout.splitlines():
        m = re.match(r"(.+?) vgpr=(\d+) agpr=(\d+) sgpr=(\d+) scratch=(\d+) occ=(\d+) vspill=(\d+)$", line.strip())
        if m:
            out[m.group(1)] = dict(zip(("vgpr", "agpr", "sgpr", "scratch", "occ", "vspill"), map(int, m.groups()[1:])))
    assert len(out) > 100, r.stdout[-2000:]
    return out


def test_no_kernel_uses_scratch(resources):
    bad = {k: v for k, v in resources.items() if v["scratch"] or v["vspill"]}
    assert not bad, bad


def test_single_pass_per_tile_kernels_keep_two_workgroups_per_cu(resources):
    # tdnn_gemm_kernel_v2<bf16 | fp16, *>: 512 threads, two workgroups per CU = four waves per SIMD = at most 128 registers
    names = [k for k in resources if re.match(r"tdnn_gemm_kernel_v2<[12], \d>", k)]
    assert len(names) == 6, names
    for k in names:
        assert resources[k]["vgpr"] <= 128 and resources[k]["occ"] >= 4, (k, resources[k])


def test_persistent_kernels_fit_one_workgroup_per_cu(resources):
    # the stream-K and first-layer kernels run 512 threads per CU: two waves per SIMD, 256 registers
    names = [k for k in resources if k.startswith("tdnn_gemm_kernel_sk<") or k.startswith("tdnn_first_kernel<") or
             k.startswith("tdnn_gemm_kernel_p8<")]
    # fp16 and fp16mx (act, stats), fp16mx2 (act, stats), fp16mxe (act)
    assert names and sum(k.startswith("tdnn_gemm_kernel_p8<") for k in names) == 7, names
    for k in names:
        assert resources[k]["vgpr"] + resources[k]["agpr"] <= 256 and resources[k]["occ"] >= 2, (k, resources[k])
