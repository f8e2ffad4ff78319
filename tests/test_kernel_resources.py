"""CPU test (-m "not gpu"): the register / scratch budgets of every gfx950 kernel in libl2d_hip.so, read from the code objects'
metadata (no GPU needed).  A kernel that starts spilling, or that outgrows the VGPR budget its launch geometry assumes, loses
occupancy silently -- the numbers in DESIGN.md (blocks per CU, waves per SIMD) are only true while these hold."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("ROCm LLVM tools not found")
    d = tmp_path_factory.mktemp("co")
    so = shutil.copy(os.path.join(ROOT, "live2diff_amd", "libl2d_hip.so"), d)
    subprocess.run([objdump, "--offloading", so], check=True, capture_output=True)           # writes the bundles next to `so`
    out = {}
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            out[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                         for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
            out[name]["agpr_count"] = int(re.match(r"\s*(\d+)", blk).group(1))
    assert len(out) > 100, f"only {len(out)} kernels found in the code objects"
    return out


def test_no_kernel_spills_or_uses_scratch(kernels):
    # known and accepted: the round-1 register-staged flash kernel at d = 160 (variant 1, kept for A/B only; the default ring
    # kernel covers d = 160) needs 100 bytes of scratch
    legacy = {"_Z20flash_attn_kernel_w2ILi160EEv6FAArgs"}
    # the runtime-K form of the token-resident GEMM (template SK = 0: unit-test widths and K = 768 only -- the SD-1.5 widths
    # 320 / 640 / 1280 take the straight-line SK = 20 / 40 / 80 forms, which must be clean): 68 bytes of scratch, no VGPR spills
    generic_k = re.compile(r"rowgemm_kernelILi\dELi\dELi\d+ELi0ELi\d+E")
    bad = {}
    for n, k in kernels.items():
        if n in legacy or not (k["vgpr_spill_count"] or k["private_segment_fixed_size"]):
            continue
        if generic_k.search(n) and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] <= 128:
            continue
        bad[n] = k
    assert not bad, bad


def test_occupancy_budgets(kernels):
    def pick(pat):
        r = {n: k for n, k in kernels.items() if re.search(pat, n)}
        assert r, pat
        return r
    # 128x128 tile with a BK = 32 ring: three (NS = 3, 4, 6) or four (NS = 2) blocks of four waves per CU -> <= 168 / 128 registers
    for n, k in pick(r"igemm_kernelILi128ELi128ELi\dELi32ELi[346]E").items():
        assert k["vgpr_count"] + k["agpr_count"] <= 168, (n, k)
    for n, k in pick(r"igemm_kernelILi128ELi128ELi\dELi32ELi2E").items():
        assert k["vgpr_count"] + k["agpr_count"] <= 128, (n, k)
    # 64x64 tile: three blocks per CU by LDS (48 KB rings); registers must not be what limits it
    for n, k in pick(r"igemm_kernelILi64ELi64ELi\dELi64ELi3E").items():
        assert k["vgpr_count"] + k["agpr_count"] <= 168, (n, k)
    # flash attention, 32 query rows per wave: two waves per SIMD; 16 rows at d <= 80: three
    for n, k in pick(r"flash_ring_kernelILi(40|80)ELi2ELi4ELi0E").items():
        assert k["vgpr_count"] + k["agpr_count"] <= 256, (n, k)
    for n, k in pick(r"flash_ring_kernelILi(40|80)ELi1ELi4ELi0E").items():
        assert k["vgpr_count"] + k["agpr_count"] <= 168, (n, k)
    # KV-cache ring kernel: 5 waves per block, one block per CU (10 waves with the 16-pixel geometry: three on two SIMDs)
    # round-2 ring form (A/B only; L = 24: one block of 5 waves per CU = at most two waves per SIMD, so up to 256 registers cost nothing)
    for n, k in pick(r"tattn_stream_ring_kernelILi\d+ELi(12|16)E").items():
        assert k["vgpr_count"] + k["agpr_count"] <= 168, (n, k)
    for n, k in pick(r"tattn_stream_ring_kernelILi\d+ELi24E").items():
        assert k["vgpr_count"] + k["agpr_count"] <= 256, (n, k)
    # loader-wave kernel (the default for every window, scores in LDS): no scratch -- the unrolled form's score array went to
    # scratch at L = 40, and a scratch load would also break the loader's counted vmcnt -- and <= 128 registers
    lw = pick(r"tattn_stream_ringlw_kernel")
    assert len(lw) == 12, sorted(lw)                       # L in {12, 16, 24, 40} x head widths {5, 10, 20} threads
    for n, k in lw.items():
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_count"] + k["agpr_count"] <= 128, (n, k)
    # token-resident GEMM, compile-time K: the geometries with up to 8 waves per block rely on <= 256 registers, no spills
    for n, k in pick(r"rowgemm_kernelILi\dELi\dELi\d+ELi(20|40|80)E").items():
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (n, k)
