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


def test_every_lds_dma_of_the_p8_kernel_sets_m0_itself(tmp_path):
    """tdnn_gemm_kernel_p8 issues its staging units through an asm form that writes m0 and does not restore it (three scalar
    instructions less per LDS-DMA in the part of a phase its barrier intervals wait for).  That is sound only while nothing in the
    kernel reads an m0 it did not just write: in the ISA of every p8 instantiation each global_load_lds is preceded, within three
    instructions, by an s_mov_b32 m0, and m0 is read by nothing else but the save / restore pairs of the other asm forms."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(H.ROOT, H.PKG_NAME, "csrc", "kernels.hip")
    out = tmp_path / "kernels.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", src, "-o", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    body, name, kernels = [], None, {}
    for line in open(out):
        m = re.match(r"^(_ZN2xv19tdnn_gemm_kernel_p8\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name and line.startswith(".Lfunc_end"):
            kernels[name] = body
            name = None
            continue
        if name:
            t = line.strip()
            if t and not t.startswith((";", ".")) and not t.endswith(":"):
                body.append(t.split(";")[0].strip())
    assert len(kernels) == 7, sorted(kernels)
    for k, ins in kernels.items():
        n_dma = 0
        for i, t in enumerate(ins):
            if t.startswith("global_load_lds"):
                n_dma += 1
                assert any(x.startswith("s_mov_b32 m0,") for x in ins[max(0, i - 3):i]), (k, ins[max(0, i - 4):i + 1])
            elif re.search(r"\bm0\b", t) and not t.startswith("s_mov_b32 m0,"):
                # the only other readers: "s_mov_b32 sN, m0" of the save / restore forms (4-byte pieces, 4-bit tiles)
                assert re.match(r"s_mov_b32 s\d+, m0$", t), (k, t)
        assert n_dma >= 16, (k, n_dma)
