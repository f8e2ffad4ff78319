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
    out["__code_objects__"] = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "gfx950" in f]
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
        if n.startswith("__") or n in legacy or not (k["vgpr_spill_count"] or k["private_segment_fixed_size"]):
            continue
        if generic_k.search(n) and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] <= 128:
            continue
        bad[n] = k
    assert not bad, bad


def test_occupancy_budgets(kernels):
    def pick(pat):
        r = {n: k for n, k in kernels.items() if not n.startswith("__") and re.search(pat, n)}
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
    # KV-cache attention: the round-2 ring form lives in analysis builds only since round 4
    assert not [n for n in kernels if "tattn_stream_ring_kernel" in n]
    # loader-wave kernel (the default for every window, scores in LDS): no scratch -- the unrolled form's score array went to
    # scratch at L = 40, and a scratch load would also break the loader's counted vmcnt -- and <= 128 registers
    lw = pick(r"tattn_stream_ringlw_kernel")
    assert len(lw) == 12, sorted(lw)                       # L in {12, 16, 24, 40} x head widths {5, 10, 20} threads
    for n, k in lw.items():
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_count"] + k["agpr_count"] <= 128, (n, k)
    # token-resident GEMM, compile-time K: the geometries with up to 8 waves per block rely on <= 256 registers, no spills
    for n, k in pick(r"rowgemm_kernelILi\dELi\dELi\d+ELi(20|40|80)E").items():
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (n, k)
    # weight-streaming GEMM: 5-6 waves per block share a CU two per SIMD (<= 256 registers), the 8-consumer form three (<= 168)
    ws = pick(r"wsgemm_kernel")
    assert len(ws) == 16, sorted(ws)          # (NT, RDS, MAXW) = (1, 4, 6), (1, 2, 10), (1, 2, 12), (2, 2, 6) x NL 1 | 2 x temporal / non-temporal loads
    for n, k in ws.items():
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (n, k)
        # (the 8-consumer form: 10 waves, the 10-consumer form of round 6: 12 waves -- three per SIMD either way)
        assert k["vgpr_count"] + k["agpr_count"] <= (168 if ("Li10EEv" in n or "Li12EEv" in n) else 256), (n, k)


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def test_wsgemm_weight_ring_registers_are_untouched_in_flight(kernels):
    """wsgemm.hip issues its weight-fragment loads and their counted waits as inline asm (the compiler's own waitcnt insertion
    would drain them at every loop back-edge).  The compiler therefore believes a ring register holds its value from the moment
    the load is ISSUED; this is only sound if nothing reads or writes such a register while its load is in flight (no copies at
    loop back-edges, no spills, no early reuse).  Checked on the shipped ISA by walking the control-flow graph of the consumer
    code of every wsgemm kernel (both outcomes of every conditional branch, states memoised so loops close) with the hardware's
    in-order VMEM return rule: a fragment load enters a FIFO, `s_waitcnt vmcnt(N)` retires all but the youngest N; no instruction
    may touch a register of a load that is still in the FIFO -- in particular every MFMA must find its A operand landed."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    n_checked = 0
    for co in kernels["__code_objects__"]:
        dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        if "wsgemm_kernel" not in dis:
            continue
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            head = fn.split("\n", 1)[0]
            if "wsgemm_kernel" not in head:
                continue
            ins, addr = [], []
            for line in fn.split("\n")[1:]:
                m = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m:
                    ins.append(m.group(1))
                    addr.append(int(m.group(2), 16))
            index = {a: i for i, a in enumerate(addr)}
            is_frag = lambda t: re.match(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[", t) is not None
            start = next(i for i, t in enumerate(ins) if is_frag(t))
            work, seen, max_depth, n_steps = [(start, ())], set(), 0, 0
            while work:
                pc, fifo = work.pop()
                while True:
                    key = (pc, fifo)
                    if key in seen or pc >= len(ins):
                        break
                    seen.add(key)
                    n_steps += 1
                    assert n_steps < 2_000_000, head
                    t = ins[pc]
                    toks = [x.rstrip(",") for x in t.split()[1:]]
                    if t.startswith("s_endpgm"):
                        break
                    if is_frag(t):
                        dst = frozenset(_regs(toks[0]))
                        inflight = set().union(*fifo) if fifo else set()
                        assert not (dst & inflight) and not (_regs(toks[1]) & inflight), f"{head}: `{t}` while in flight"
                        fifo = fifo + (dst,)
                        max_depth = max(max_depth, len(fifo))
                        pc += 1
                        continue
                    m = re.search(r"vmcnt\((\d+)\)", t)
                    if t.startswith("s_waitcnt") and m:
                        fifo = fifo[max(0, len(fifo) - int(m.group(1))):]
                        pc += 1
                        continue
                    if t.startswith("s_branch") or t.startswith("s_cbranch"):
                        imm = int(toks[0])
                        tgt = index[addr[pc] + 4 + 4 * (imm - 65536 if imm >= 32768 else imm)]
                        if t.startswith("s_cbranch"):
                            work.append((pc + 1, fifo))
                        pc = tgt
                        continue
                    if fifo:
                        assert not re.match(r"(global_|buffer_|flat_|scratch_)", t), f"{head}: `{t}`: foreign VMEM op inside the counted region"
                        touched = set().union(*[_regs(x) for x in toks]) if toks else set()
                        inflight = set().union(*fifo)
                        assert not (touched & inflight), f"{head}: `{t}` touches a fragment register whose load is still in flight"
                    pc += 1
            assert max_depth in (8, 16), (head, max_depth)
            n_checked += 1
    assert n_checked == 16, n_checked


def test_wsgemm_lds_fragment_registers_are_untouched_in_flight(kernels):
    """The same replay for the LDS queue: wsgemm.hip reads its activation fragments with inline-asm ds_read_b128 and waits with
    counted lgkmcnt (LDS returns in order).  Walks every path of every wsgemm kernel from its entry; a ds_read enters a FIFO,
    `s_waitcnt lgkmcnt(N)` retires all but the youngest N; no instruction may read or write a register of a read that is still in
    the FIFO.  (Round 4 shipped a build, for an hour, in which hipcc copied a fragment register at a control-flow merge before its
    read had been waited for: bit-exact alone, wrong rows under a concurrent LDS-heavy kernel.)"""
    objdump = os.path.join(LLVM, "llvm-objdump")
    n_checked = 0
    for co in kernels["__code_objects__"]:
        dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        if "wsgemm_kernel" not in dis:
            continue
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            head = fn.split("\n", 1)[0]
            if "wsgemm_kernel" not in head:
                continue
            ins, addr = [], []
            for line in fn.split("\n")[1:]:
                m = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m:
                    ins.append(m.group(1))
                    addr.append(int(m.group(2), 16))
            index = {a: i for i, a in enumerate(addr)}
            work, seen, max_depth, n_steps = [(0, ())], set(), 0, 0
            while work:
                pc, fifo = work.pop()
                while True:
                    key = (pc, fifo)
                    if key in seen or pc >= len(ins):
                        break
                    seen.add(key)
                    n_steps += 1
                    assert n_steps < 4_000_000, head
                    t = ins[pc]
                    toks = [x.rstrip(",") for x in t.split()[1:]]
                    if t.startswith("s_endpgm"):
                        break
                    if t.startswith("ds_read"):
                        dst = frozenset(_regs(toks[0]))
                        inflight = set().union(*fifo) if fifo else set()
                        assert not (dst & inflight) and not (_regs(toks[1]) & inflight), f"{head}: `{t}` on registers of an LDS read in flight"
                        fifo = fifo + (dst,)
                        max_depth = max(max_depth, len(fifo))
                        pc += 1
                        continue
                    m = re.search(r"lgkmcnt\((\d+)\)", t)
                    if t.startswith("s_waitcnt") and m:
                        fifo = fifo[max(0, len(fifo) - int(m.group(1))):]
                        pc += 1
                        continue
                    if t.startswith("s_branch") or t.startswith("s_cbranch"):
                        imm = int(toks[0])
                        tgt = index[addr[pc] + 4 + 4 * (imm - 65536 if imm >= 32768 else imm)]
                        if t.startswith("s_cbranch"):
                            work.append((pc + 1, fifo))
                        pc = tgt
                        continue
                    if fifo:
                        touched = set().union(*[_regs(x) for x in toks]) if toks else set()
                        inflight = set().union(*fifo)
                        assert not (touched & inflight), f"{head}: `{t}` touches a fragment register whose LDS read is still in flight"
                    pc += 1
            assert max_depth >= 8, (head, max_depth)
            n_checked += 1
    assert n_checked == 16, n_checked


def test_no_kernel_uses_packed_fp32_math(kernels):
    """The library is compiled with the packed-fp32 feature off (csrc/Makefile): with v_pk_mul_f32 / v_pk_fma_f32 in wsgemm's
    LayerNorm-fold epilogue the kernel's output depended -- only while another kernel shared the GPU -- on the machine's load (the
    colsum product of ONE instruction came out 0 in its last 16 lanes; round 4, tools/wsgemm_diag.py, tools/frame_stress.py), and the
    arithmetic reproduces in isolation (tools/pk_repro.py).  The flag is one line in a Makefile; this keeps it from getting lost."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    seen_ws = 0
    for co in kernels["__code_objects__"]:
        dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            head = fn.split("\n", 1)[0]
            if "wsgemm_kernel" in head:
                seen_ws += 1
            bad = sorted(set(re.findall(r"\bv_pk_\w+_f32\b", fn)))
            assert not bad, (head, bad)
    assert seen_ws == 16, seen_ws


def test_skinny_linear_rows_share_one_instruction_sequence(kernels):
    """Root cause of round 4's red GPU suite (test_cfg2_stream_batch_rows_are_independent, rel 2.06e-3 on an idle GPU): the rows of
    the time-embedding GEMV are unrolled inside one thread, and hipcc gave row 0 a v_fma_mix_f32 chain and row 1 v_dot2c_f32_f16
    once packed fp32 math was switched off -- two roundings of the same sum, so the stream-batch rows were no longer
    interchangeable.  The kernel now spells the dot product out (misc.hip); this asserts every unrolled row got the same instructions:
    4 v_dot2(c)_f32_f16 per row and 16-byte column piece, no other fp32 multiply-add in the loop."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    seen = {}
    for co in kernels["__code_objects__"]:
        dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            head = fn.split("\n", 1)[0]
            m = re.search(r"skinny_linear_kernelILi(\d)E", head)
            if not m or ".kd>" in head:          # (the kernel descriptors are symbols too)
                continue
            mm = int(m.group(1))
            seen[mm] = (len(re.findall(r"\bv_dot2c?_f32_f16", fn)), sorted(set(re.findall(r"\bv_(?:fma_mix|mad_mix|fma|mac|fmac|mad)_\w*f(?:32|16)\w*", fn))))
    assert sorted(seen) == list(range(1, 9)), seen
    for mm, (ndot, other) in seen.items():
        assert ndot == 4 * mm and not other, (mm, ndot, other)


def test_cconv_chunk_loop_is_counted_not_drained(kernels):
    """cconv.hip streams its weights through a register ring with PLAIN loads and relies on the compiler's waitcnt insertion to count
    them across the rolled chunk loop (`s_waitcnt vmcnt(16..18)`: the ring stays in flight).  Two things broke that during bring-up and
    are pinned here on the shipped ISA: (a) with the LDS-DMA builtin anywhere in the kernel the loop head waited vmcnt(0) -- the DMA is
    inline asm now and must stay invisible to the compiler; (b) control flow inside the loop body spilled the ring.  For every cconv
    kernel: the hot loop (the basic block run with >= 72 MFMAs and a backward branch) holds no `vmcnt(0)`, at least one counted vmcnt
    wait per k step pair, exactly the expected MFMAs / fragment reads / weight loads, and the kernel uses no scratch."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    n_checked = 0
    for co in kernels["__code_objects__"]:
        dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
        if "cconv_kernel" not in dis:
            continue
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", dis):
            head = fn.split("\n", 1)[0]
            if "cconv_kernel" not in head:
                continue
            ins, addr = [], []
            for line in fn.split("\n")[1:]:
                m = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m:
                    ins.append(m.group(1))
                    addr.append(int(m.group(2), 16))
            # backward branches delimit loops: take the one whose body holds the most MFMAs
            best = None
            for i, t in enumerate(ins):
                m = re.match(r"s_cbranch_\w+ (\d+)", t)
                if not m:
                    continue
                off = int(m.group(1))
                if off < 32768:
                    continue                                    # forward branch
                tgt = addr[i] + 4 + 4 * (off - 65536)
                if tgt in addr:
                    j = addr.index(tgt)
                    body = ins[j:i + 1]
                    inner = [re.match(r"s_cbranch_\w+ (\d+)", b) for b in body[:-1]]
                    if any(m_ and int(m_.group(1)) >= 32768 for m_ in inner):
                        continue                                # not an innermost loop
                    nm = sum("v_mfma_f32_32x32x16_f16" in b for b in body)
                    if best is None or nm > best[0]:
                        best = (nm, body)
            assert best is not None and best[0] == 144, (head, None if best is None else best[0])      # one chunk (KG = 2) or two (KG = 4)
            body = best[1]
            waits = [b for b in body if b.startswith("s_waitcnt") and "vmcnt" in b]
            assert waits and not any(re.search(r"vmcnt\(0\)", w) for w in waits), (head, waits[:4])
            assert all(int(re.search(r"vmcnt\((\d+)\)", w).group(1)) >= 14 for w in waits), (head, waits[:6])
            assert sum(b.startswith("ds_read_b128") for b in body) == 72 and sum(b.startswith("global_load_dwordx4") for b in body) == 36, head
            assert not any("scratch_" in b for b in ins) and not any("global_load_lds" in b and "ASM" in b for b in body), head
            n_checked += 1
    assert n_checked >= 6, n_checked
    for n, k in kernels.items():
        if "cconv_kernel" in n:
            assert k["vgpr_count"] <= 256 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (n, k)
