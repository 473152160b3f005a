"""Compile-time guard on the hot kernels (no GPU needed: hipcc cross-compiles gfx950 here).

Round 6 found out the hard way what a private array in scratch memory costs: a refactor of the big-tile NT kernel left the
compiler indexing the MFMA accumulators dynamically in ONE epilogue instantiation (the split-K FFN2 + residual GEMM), the
accumulator array moved to scratch, every parity test stayed green and the kernel ran 6x slower (294 instead of 47 us; step
7.4 -> 11.4 ms).  `-Rpass-analysis=kernel-resource-usage` reports it at build time, so this test compiles the two kernel
files that hold every MFMA kernel of the path and requires ScratchSize = 0 (no private-memory traffic, no spills) for each of
them, and the register budget of the one-workgroup-per-CU kernels (<= 256 VGPRs = two waves per SIMD).
"""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mint_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernels allowed a few bytes of scratch: debug-only variants that no default path launches
ALLOW_SCRATCH = {
    # 384x192 tile on 32x32x16 MFMAs, fp32 + residual epilogue (nt_variant 23, measured slower than the 16x16x32 tiles and not
    # selected by the dispatcher): one spilled VGPR in the epilogue
    "big_nt_kernelINS_6BigCfgILi4ELi3ELi2ELi3ELi2ELi1ELi64ELi32EEELi3EE": 16,
    # 256x256 on 32x32x16 MFMAs, fp32 epilogues (nt_variant 22, same status)
    "big_nt_kernelINS_6BigCfgILi2ELi4ELi4ELi2ELi2ELi1ELi64ELi32EEELi1EE": 128,
    "big_nt_kernelINS_6BigCfgILi2ELi4ELi4ELi2ELi2ELi1ELi64ELi32EEELi3EE": 128,
}


def _resource_usage(src, tmp):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                          "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, src + ".o")],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        for key in ("ScratchSize [bytes/lane]", "VGPRs", "VGPRs Spill", "SGPRs Spill", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    return kernels


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    tmp = str(tmp_path_factory.mktemp("kres"))
    with ThreadPoolExecutor(2) as ex:
        a, b = ex.map(lambda f: _resource_usage(f, tmp), ["gemm_big.hip", "attention.hip"])
    return {"gemm_big.hip": a, "attention.hip": b}


def test_no_mfma_kernel_touches_scratch_memory(usage):
    bad = []
    for src, kernels in usage.items():
        assert len(kernels) > 10, (src, len(kernels))
        for name, u in kernels.items():
            allowed = max([v for k, v in ALLOW_SCRATCH.items() if k in name] or [0])
            if u.get("ScratchSize [bytes/lane]", 0) > allowed or (allowed == 0 and u.get("VGPRs Spill", 0) > 0):
                bad.append((src, name, u))
    assert not bad, "kernels with private-memory (scratch) traffic:\n" + "\n".join(map(str, bad))


def test_register_budget_of_the_one_per_cu_kernels(usage):
    """512-thread workgroups put two waves on each SIMD: <= 256 registers per lane, or the workgroup does not launch."""
    for name, u in usage["gemm_big.hip"].items():
        if "big_nt_kernel" in name or "big_tn_kernel" in name:
            assert u["VGPRs"] <= 256, (name, u)
    # the shipped attention kernels: lean backward at 12 / 8 waves per workgroup, streaming forward at 3 workgroups per CU
    att = usage["attention.hip"]
    dq = [u for n, u in att.items() if "attn_bwd_dq_r2_kernelILi80" in n]
    dkdv = [u for n, u in att.items() if "attn_bwd_dkdv_r2_kernelILi80" in n]
    fwd = [u for n, u in att.items() if "attn_fwd_st_kernelILi80" in n]
    assert dq and dkdv and fwd
    assert all(u["VGPRs"] <= 168 for u in dq + fwd), (dq, fwd)     # 3 waves per SIMD
    assert all(u["VGPRs"] <= 256 for u in dkdv), dkdv             # 2 waves per SIMD
