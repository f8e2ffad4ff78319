"""Builders for `l2d_op` records (include/l2d.h) from torch device tensors, plus the one-time weight
packers.  Everything here is plumbing: pointers, sizes and strides; the arithmetic is in csrc/*.hip.

Each builder returns `(op, keepalive_tensors)` so that a plan (`_lib.OpList`) can keep the buffers alive.
"""
import os
from typing import Optional

import torch

from . import _lib
from ._lib import L2dOp


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


DRY_RUN = False   # set by the CPU test-suite together with l2d_set_dry_run(1): plans are built and validated, never launched


def _h(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.float16 and (t.is_cuda or DRY_RUN), (t.dtype, t.device)
    return t


# ----------------------------------------------------------------------------- weight packing (one-time)
def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [Cout, 9*CinP] fp16 with k = tap*CinP + ci, CinP = round_up(Cin, 64) (zero padded)."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cinp = round_up(cin, 64)
    out = torch.zeros(cout, 9, cinp, dtype=torch.float16, device=w.device)
    out[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin).to(torch.float16)
    return out.reshape(cout, 9 * cinp).contiguous()


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    """[Nout,K] (or [Nout,K,1,1]) -> [Nout, round_up(K,64)] fp16, zero padded."""
    w = w.reshape(w.shape[0], -1)
    n, k = w.shape
    kp = round_up(k, 64)
    out = torch.zeros(n, kp, dtype=torch.float16, device=w.device)
    out[:, :k] = w.to(torch.float16)
    return out.contiguous()


def geglu_perm(c4: int, device) -> torch.Tensor:
    """Row permutation for the GEGLU projection [8C, C]: packed rows are 16 value rows then the 16 matching
    gate rows, so that value and gate of one output column meet in the same MFMA lane (igemm.hip epilogue)."""
    assert c4 % 16 == 0
    blk = torch.arange(c4 // 16, device=device)[:, None] * 16 + torch.arange(16, device=device)[None]
    return torch.stack([blk, blk + c4], dim=1).reshape(-1)


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    perm = geglu_perm(w.shape[0] // 2, w.device)
    return pack_linear(w[perm]), b[perm].float().contiguous()


def f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.float().contiguous()


ROWGEMM_MAX_K = 2048


def rowgemm_ok(nout: int, k: int) -> bool:
    """Shapes the token-row GEMM (csrc/rowgemm.hip) takes: whole K resident in LDS, 32-row weight tiles."""
    return k % 64 == 0 and 64 <= k <= ROWGEMM_MAX_K and nout % 32 == 0 and nout >= 32


def rowgemm_geglu_perm(c4: int, device) -> torch.Tensor:
    """Row order of a GEGLU projection [8C -> value | gate] for the 32x32 MFMA tile of rowgemm.hip: a 32-row tile holds 8
    value rows, the 8 matching gate rows, the next 8 value rows and their gate rows, so that value and gate of one output
    channel are the register groups (0, 1) / (2, 3) of the same lane."""
    assert c4 % 16 == 0
    t = torch.arange(c4 // 16, device=device)[:, None, None] * 16            # tile -> first output channel
    h = torch.arange(2, device=device)[None, :, None] * 8                    # half of the tile
    e = torch.arange(8, device=device)[None, None, :]
    val = (t + h + e)                                                        # [tiles, 2, 8]
    return torch.stack([val, val + c4], dim=2).reshape(-1)                   # [tiles, 2, (value, gate), 8]


def pack_rowgemm(w: torch.Tensor, bias: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
                 beta: Optional[torch.Tensor] = None, geglu: bool = False):
    """[Nout, K] (or [Nout, K, 1, 1]) -> (weights in MFMA-fragment order, fp32 bias or None) for rowgemm.hip.

    Fragment order: flat index (((t * S + s) * 64 + lane) * 8 + e) holds W[32 t + lane % 32][16 s + 8 (lane // 32) + e],
    S = K / 16: the A operand of v_mfma_f32_32x32x16_f16 for weight tile t and k step s is one contiguous 1 KB block.
    With `gamma` / `beta` (the affine parameters of the LayerNorm / GroupNorm in front of this layer) the affine map is
    folded into the layer: W' = W diag(gamma), b' = b + W beta (computed in fp32 from the fp16 parameters), so the kernel's
    prologue only normalises.  `geglu`: rows re-ordered by rowgemm_geglu_perm (bias likewise)."""
    w = w.reshape(w.shape[0], -1).float()
    n, k = w.shape
    assert rowgemm_ok(n, k), (n, k)
    b = None if bias is None else bias.float().clone()
    if gamma is not None:
        if beta is not None:
            shift = w @ beta.float()
            b = shift if b is None else b + shift
        w = w * gamma.float()[None, :]
    if geglu:
        perm = rowgemm_geglu_perm(n // 2, w.device)
        w = w[perm]
        b = None if b is None else b[perm]
    S = k // 16
    packed = w.to(torch.float16).view(n // 32, 32, S, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    return packed, (None if b is None else b.contiguous())


def unpack_rowgemm(packed: torch.Tensor, nout: int, k: int) -> torch.Tensor:
    """inverse of the fragment permutation (tests)"""
    S = k // 16
    return packed.view(nout // 32, S, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(nout, k)


# ----------------------------------------------------------------------------- op builders
_ZERO_PAGES = {}


def zero_page(device) -> torch.Tensor:
    """64 KB of zeros per device: the DMA source for conv padding / ragged rows / channel tails (igemm.hip reads 16 bytes of it;
    wsgemm.hip walks up to 2 CinP + 128 bytes of it with the same per-stage stride as real rows)."""
    key = str(device)
    if key not in _ZERO_PAGES:
        _ZERO_PAGES[key] = torch.zeros(32768, dtype=torch.float16, device=device)
    return _ZERO_PAGES[key]


def igemm_gn_target(op, acc_ptr: int, *, T: int, G: int, cpg: int, choff: int) -> bool:
    """Ask an igemm op to accumulate the GroupNorm statistics of its output for one consumer GroupNorm (up to two per op):
    `acc_ptr` -> int64 [samples][G][2], cpg / choff = channels per group and channel offset of this tensor in the consumer's
    (concatenated) channel axis.  Returns False when both slots are taken or the launch cannot do it (include/l2d.h)."""
    assert op.kind == _lib.OP_IGEMM
    splitk, tile = max(1, op.i[21]), op.i[22] & 15
    fused = bool(op.p[11])
    tm = 64 if ((splitk > 1 and not fused) or tile == 2) else 128
    Nout, ldo, ldr, epi, batch = op.i[14], op.i[15], op.i[16], op.i[19], max(1, op.i[20])
    direct = (op.i[22] >> 5) & 1
    vec_ok = ((not direct) and Nout % 8 == 0 and ldo % 8 == 0 and not (op.p[5] and ldr % 8) and epi != 1
              and (op.p[6] or 0) % 16 == 0 and (op.p[5] or 0) % 16 == 0)     # 16-byte aligned out / residual (sub-views)
    if T % tm or batch != 1 or ((splitk == 1 or fused) and not vec_ok) or op.i[13] % T or G > 32:
        return False
    if op.p[9] and (op.i[24], op.i[25]) != (T, G):
        return False
    slot = 0 if not op.p[9] else (1 if not op.p[10] else -1)
    if slot < 0:
        return False
    op.p[9 + slot] = int(acc_ptr)
    op.i[24], op.i[25] = int(T), int(G)
    op.i[26 + 2 * slot], op.i[27 + 2 * slot] = int(cpg), int(choff)
    return True


def rowgemm_gn_target(op, acc_ptr: int, *, T: int, G: int, cpg: int, choff: int) -> bool:
    """The same request to a rowgemm op (its epilogue accumulates like igemm's LDS-staged one)."""
    assert op.kind == _lib.OP_ROWGEMM
    MT, ntr, epi = op.i[14], op.i[15], op.i[6]
    if T % (32 * MT) or op.i[0] % T or ntr or epi == 1 or G > 32 or (cpg | choff) & 1:
        return False
    if op.p[9] and (op.i[24], op.i[25]) != (T, G):
        return False
    slot = 0 if not op.p[9] else (1 if not op.p[10] else -1)
    if slot < 0:
        return False
    op.p[9 + slot] = int(acc_ptr)
    op.i[24], op.i[25] = int(T), int(G)
    op.i[26 + 2 * slot], op.i[27 + 2 * slot] = int(cpg), int(choff)
    return True


def pconv_gn_target(op, acc_ptr: int, *, T: int, G: int, cpg: int, choff: int) -> bool:
    """The same request to a patch-conv op (a patch never straddles samples)."""
    assert op.kind == _lib.OP_PCONV
    if T != op.i[7] * op.i[8] or G > 32 or (cpg | choff) & 1:
        return False
    if op.p[9] and (op.i[24], op.i[25]) != (T, G):
        return False
    slot = 0 if not op.p[9] else (1 if not op.p[10] else -1)
    if slot < 0:
        return False
    op.p[9 + slot] = int(acc_ptr)
    op.i[24], op.i[25] = int(T), int(G)
    op.i[26 + 2 * slot], op.i[27 + 2 * slot] = int(cpg), int(choff)
    return True


def gn_target(op, acc_ptr: int, **kw) -> bool:
    """Ask the GEMM op that produced a tensor to accumulate that tensor's GroupNorm statistics (igemm, rowgemm or pconv)."""
    if op.kind == _lib.OP_IGEMM:
        return igemm_gn_target(op, acc_ptr, **kw)
    if op.kind == _lib.OP_ROWGEMM:
        return rowgemm_gn_target(op, acc_ptr, **kw)
    if op.kind == _lib.OP_PCONV:
        return pconv_gn_target(op, acc_ptr, **kw)
    if op.kind == _lib.OP_WSGEMM:
        return wsgemm_gn_target(op, acc_ptr, **kw)
    if op.kind == _lib.OP_ROWCHAIN:
        return rowchain_gn_target(op, acc_ptr, **kw)
    if op.kind == _lib.OP_CCONV:
        return cconv_gn_target(op, acc_ptr, **kw)
    return False


def rowchain_gn_target(op, acc_ptr: int, *, T: int, G: int, cpg: int, choff: int) -> bool:
    """The same request to a rowchain op (its row phase accumulates like the row GEMM's; a block is 32 tokens of one sample)."""
    assert op.kind == _lib.OP_ROWCHAIN
    if op.i[6] != 0 or T % 32 or op.i[0] % T or G > 32 or (cpg | choff) & 1:          # (head segments feed attention, never a GroupNorm)
        return False
    if op.p[9] and (op.i[24], op.i[25]) != (T, G):
        return False
    slot = 0 if not op.p[9] else (1 if not op.p[10] else -1)
    if slot < 0:
        return False
    op.p[9 + slot] = int(acc_ptr)
    op.i[24], op.i[25] = int(T), int(G)
    op.i[26 + 2 * slot], op.i[27 + 2 * slot] = int(cpg), int(choff)
    return True


def rowchain_head(x, hout, out, *, M, C, wA, bA, wB, bB=None, passes=1, resA=None, gn_acc_ptr=None, T=0, G=0, eps_gn=1e-6, eps_ln=1e-5,
                  out_t=None, ldt=0, st=0, ldx=None, ldrA=None, ldh=None, ldo=None):
    """Head segment of a transformer block (csrc/rowchain.hip, two dependent layers in one launch):
    h = A(norm?(x)) + bA (+ resA) -> hout;  out = B(LayerNorm(h)) + bB, `passes` x C packed rows; with `out_t` the LAST pass is stored
    transposed (V^T[sample][channel][ldt]) and `out` receives the first passes - 1.  gn_acc_ptr: GroupNorm prologue on x (statistics
    [samples][G][2] from x's producers, T tokens per sample).  Weights: pack_rowgemm forms (norm affine folded)."""
    op = L2dOp()
    op.kind = _lib.OP_ROWCHAIN
    assert wA.dtype == torch.float16 and wA.numel() == C * C and wB.dtype == torch.float16 and wB.numel() == passes * C * C
    assert bA.dtype == torch.float32 and bA.numel() == C and (bB is None or (bB.dtype == torch.float32 and bB.numel() == passes * C))
    trl = out_t is not None
    ncol = (passes - (1 if trl else 0)) * C
    op.p[0], op.p[1], op.p[2] = _ptr(_h(x)), (_ptr(_h(resA)) if resA is not None else None), _ptr(_h(hout))
    op.p[3] = _ptr(_h(out)) if out is not None else None
    op.p[4], op.p[5], op.p[6], op.p[7] = _ptr(_h(wA)), _ptr(bA), _ptr(_h(wB)), _ptr(bB)
    op.p[8] = _ptr(_h(out_t)) if trl else None
    op.p[14] = int(gn_acc_ptr) if gn_acc_ptr is not None else None
    op.i[0], op.i[1] = int(M), int(C)
    op.i[2], op.i[3], op.i[4], op.i[5] = int(ldx or C), int(ldrA or C), int(ldh or C), int(ldo or max(ncol, C))
    op.i[6], op.i[7], op.i[8], op.i[9], op.i[10], op.i[11] = 1, int(passes), int(trl), int(ldt), int(T), int(G)
    op.l[0] = int(st)
    op.f[0], op.f[1] = float(eps_ln), float(eps_gn)
    return op, (x, resA, hout, out, out_t, wA, bA, wB, bB)


ROWCHAIN_C = 320                # the width the chain kernel is instantiated for (SD-1.5 level 0)
# M / 32 blocks.  Round 5 set 192 from the BASELINE configs (256+ blocks, or cfg-1's 32); round 6 measured the gap: 144 blocks (384 x 384,
# N = 2) 7.70 -> 7.40 ms, 154 blocks (448 x 704, N = 1) 8.34 -> 7.98 ms, 100 blocks (320 x 320, N = 2) 7.09 -> 7.00 ms with the chain
# (profiles/round6_v_*, round6_w_*); cfg-1 (32 blocks) keeps the separate launches: 5.14 ms against 5.31 with the chain (round6_ab_*)
ROWCHAIN_MIN_BLOCKS = int(os.environ.get("L2D_ROWCHAIN_MIN_BLOCKS", "96"))


def rowchain_ok(M: int, C: int, T: int) -> bool:
    """Shapes the token-resident block tail (csrc/rowchain.hip) takes"""
    return (os.environ.get("L2D_ROWCHAIN", "1") != "0" and C == ROWCHAIN_C and M % 32 == 0 and T % 32 == 0
            and M // 32 >= ROWCHAIN_MIN_BLOCKS)


def rowchain(a, res1, res2, out, *, M, C, w_out, b_out, w_ff1, b_ff1, w_ff2, b_ff2, w_po, b_po, eps=1e-5, lda=None, ldr1=None,
             ldr2=None, ldo=None):
    """Token-resident tail of a transformer block (csrc/rowchain.hip): out = proj_out(FF2(GEGLU(LN(h2))) + h2) + res2 with
    h2 = to_out(a) + res1.  Weights / biases: pack_rowgemm forms (to_out plain; FF1 with the LayerNorm folded and geglu=True; FF2;
    proj_out)."""
    op = L2dOp()
    op.kind = _lib.OP_ROWCHAIN
    for t_, n_ in ((w_out, C * C), (w_ff1, 8 * C * C), (w_ff2, 4 * C * C), (w_po, C * C)):
        assert t_.dtype == torch.float16 and t_.numel() == n_, (t_.shape, n_)
    for t_, n_ in ((b_out, C), (b_ff1, 8 * C), (b_ff2, C), (b_po, C)):
        assert t_.dtype == torch.float32 and t_.numel() == n_, (t_.shape, n_)
    op.p[0], op.p[1], op.p[2], op.p[3] = _ptr(_h(a)), _ptr(_h(res1)), _ptr(_h(res2)), _ptr(_h(out))
    op.p[4], op.p[5], op.p[6], op.p[7] = _ptr(_h(w_out)), _ptr(b_out), _ptr(_h(w_ff1)), _ptr(b_ff1)
    op.p[8], op.p[11], op.p[12], op.p[13] = _ptr(_h(w_ff2)), _ptr(b_ff2), _ptr(_h(w_po)), _ptr(b_po)
    op.i[0], op.i[1] = int(M), int(C)
    op.i[2], op.i[3], op.i[4], op.i[5] = int(lda or C), int(ldr1 or C), int(ldr2 or C), int(ldo or C)
    op.f[0] = float(eps)
    return op, (a, res1, res2, out, w_out, b_out, w_ff1, b_ff1, w_ff2, b_ff2, w_po, b_po)


def pconv_patch(B: int, H: int, W: int, Nout: int, C1: int, C2: int = 0):
    """Patch (PH, PW) for the patch-resident 3x3 conv, or None when the implicit-GEMM kernel should take the launch: the largest
    of 8x16 / 8x8 / 4x8 that tiles the image and yields >= 256 blocks of (patch, 64 output channels).  Not for the lowest
    resolutions: with few tokens the launch is bound by its weight stream (9 C Nout x 2 bytes), which wants split-K."""
    if os.environ.get("L2D_PCONV", "1") == "0" or Nout % 64 or C1 % 64 or C2 % 64 or C1 <= 0:
        return None
    for ph, pw in ((8, 16), (8, 8), (4, 8)):
        if H % ph or W % pw:
            continue
        blocks = B * (H // ph) * (W // pw) * (Nout // 64)
        tokens = B * H * W
        if blocks >= 256 and tokens >= 2048:
            return ph, pw
    return None


def pconv(x1, w, out, *, B, H, W, C1, ldx1, CinP, Nout, ldo, patch, x2=None, C2=0, ldx2=0, bias=None, rowbias=None, ldrb=0,
          rows_per_bias=0, res=None, ldr=0, order=None, epi=0):
    """3x3 stride-1 conv, activation patch resident in LDS (csrc/pconv.hip); `w` = pack_conv3x3 weights (as igemm).
    epi as igemm: 0 none, 2 SiLU, 3 ReLU, 5 GELU, 4 ReLU after the residual add."""
    op = L2dOp()
    op.kind = _lib.OP_PCONV
    zp = zero_page(x1.device)
    op.p[0], op.p[1], op.p[2] = _ptr(_h(x1)), (_ptr(x2) if x2 is not None else None), _ptr(_h(w))
    op.p[3], op.p[4], op.p[5] = _ptr(bias), _ptr(rowbias), (_ptr(_h(res)) if res is not None else None)
    op.p[6], op.p[7] = _ptr(_h(out)), _ptr(zp)
    assert CinP == C1 + C2 and w.shape[1] == 9 * CinP and w.shape[0] == Nout
    if order is None:
        order = int(Nout * 9 * CinP > B * H * W * (C1 + C2))
    vals = {1: C1, 2: C2, 3: ldx1, 4: ldx2, 5: CinP, 6: B, 7: H, 8: W, 9: patch[0], 10: patch[1], 11: order, 14: Nout, 15: ldo,
            16: ldr, 17: ldrb, 18: rows_per_bias, 19: epi}
    for j, v in vals.items():
        op.i[j] = int(v)
    return op, (x1, x2, w, bias, rowbias, res, out, zp)



def _rowgemm_lds(K, NW, NT, MT, epi, pro, ntr, gn):
    BM, BNp = 32 * MT, NW * NT * 32
    BNo = BNp // 2 if epi == 1 else BNp
    xs = BM * K * 2 + BM * 16 + ((2 * K + 64) * 4 if pro == 2 else 0)
    os_ = BM * (BNo + 8) * 2 + (64 * NW * 32 + BNo * 4 if gn else 0)
    if ntr:
        os_ = max(os_, BNp * (BM + 8) * 2)
    return max(xs, os_)


def _rowgemm_mt_ok(mt, nt, K, T):
    """token tiles per block: 1, 2 (NT <= 2) or 4 (NT <= 2, K = 320: 80 KB of LDS); with T given (a norm prologue from group
    statistics, a transposed output or output statistics) a sample must be a whole number of token tiles"""
    if mt == 1:
        return True
    if mt not in (2, 4) or nt > 2 or (mt == 4 and K != 320):
        return False
    return T % (32 * mt) == 0


def rowgemm_schedule(M: int, K: int, Nout: int, ntr: int = 0, epi: int = 0, pro: int = 0, gn: bool = True, T: int = 0):
    """(NW, NT, MT) = waves per block, 32-row weight tiles per wave, 32-token tiles per block for a rowgemm launch.  A table
    measured on MI355X (rowgemm_tuned.json, tools/rowgemm_sweep.py) when it holds the shape, else a small cost model:
    per block one activation-tile load (dependent groups of 5 x 16 B per thread), then per wave NT * K * 64 bytes of weights
    through a ring of ~16 KB (latency bound: ~20 GB/s per wave) or NT * MT * K / 16 MFMAs of 32 cycles, whichever is longer;
    blocks run in rounds of 256 CUs x (blocks per CU by LDS and by 12 waves)."""
    tiles = Nout // 32
    force = os.environ.get("L2D_ROWGEMM_FORCE")        # "NW,NT,MT" (tools): applied where it divides the shape
    if force:
        nw, nt, mt = (int(v) for v in force.split(","))
        if tiles % (nw * nt) == 0 and (ntr // 32) % (nw * nt) == 0 and nw <= (5 if nt >= 3 else 8) and _rowgemm_mt_ok(mt, nt, K, T) \
                and _rowgemm_lds(K, nw, nt, mt, epi, pro, ntr, gn) <= 163840:
            return nw, nt, mt
    key = f"{M},{K},{Nout},{ntr},{epi}"
    if key in _RG_TUNED and _rowgemm_mt_ok(_RG_TUNED[key][2], _RG_TUNED[key][1], K, T):
        return tuple(_RG_TUNED[key])
    cdiv = lambda a, b: (a + b - 1) // b
    best, best_t = None, None
    for mt in (1, 2):
        if mt == 2 and (M < 4096 or T % 64):        # (a sample must be a whole number of token tiles for prologue 2 / V^T / statistics)
            continue
        for nt in (1, 2, 3, 4):
            if mt == 2 and nt > 2:
                continue
            for nw in range(1, (5 if nt >= 3 else 8) + 1):
                if tiles % (nw * nt) or (ntr // 32) % (nw * nt):
                    continue
                lds = _rowgemm_lds(K, nw, nt, mt, epi, pro, ntr, gn)
                if lds > 163840:
                    continue
                bm = 32 * mt
                blocks = cdiv(M, bm) * (tiles // (nw * nt))
                bpc = max(1, min(163840 // lds, 12 // nw))
                rounds = cdiv(blocks, 256 * bpc)
                lpr = 16 if (64 * nw >= 16 * bm and K % 128 == 0) else 8
                row_passes = cdiv(bm, 64 * nw // lpr)
                t_x = 0.9 * row_passes * cdiv(K // (8 * lpr), 5) + (1.0 if pro == 2 else 0.0) + (0.5 if pro == 1 else 0.0)
                t_w = nt * K * 64 / 20000.0
                t_mm = nt * mt * (K / 16) * 32 * cdiv(nw * bpc, 4) / 2100.0
                t = rounds * (t_x + max(t_w, t_mm) + 0.8)
                if best_t is None or t < best_t - 1e-9 or (abs(t - best_t) <= 1e-9 and blocks < best[3]):
                    best, best_t = (nw, nt, mt, blocks), t
    assert best is not None, (M, K, Nout, ntr)
    return best[:3]


def _load_rg_tuned():
    import json
    path = os.path.join(os.path.dirname(__file__), "rowgemm_tuned.json")
    if os.environ.get("L2D_ROWGEMM_NO_TABLE") or not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)["shapes"]


_RG_TUNED = _load_rg_tuned()


def rowgemm(x, w, out, *, M, K, Nout, ldx, ldo, bias=None, res=None, ldr=0, epi=0, pro=0, eps=1e-5, T=0, G=0, gn_acc_ptr=None,
            out_t=None, ntr=0, ldt=0, st=0, sched=None, order=None, x_off=0, out_off=0, res_off=0):
    """Token-row GEMM (csrc/rowgemm.hip): out[m][n] = epi(sum_k norm(x)[m][k] W[n][k]).  `w` / `bias` come from pack_rowgemm
    (affine of the norm folded in).  pro: 0 none, 1 LayerNorm over K, 2 GroupNorm from the fixed-point accumulators at
    `gn_acc_ptr` (T tokens per sample, G groups).  The LAST `ntr` packed rows are stored transposed into out_t
    [sample][row][ldt] (V^T).  sched = (NW, NT, MT) or None for rowgemm_schedule."""
    op = L2dOp()
    op.kind = _lib.OP_ROWGEMM
    assert w.dtype == torch.float16 and w.numel() == Nout * K, (w.shape, Nout, K)
    if sched is None:
        sched = rowgemm_schedule(M, K, Nout, ntr, epi, pro, T=T)
    NW, NT, MT = sched
    op.p[0] = _ptr(_h(x)) + 2 * x_off
    op.p[1] = _ptr(_h(w))
    op.p[2] = _ptr(bias)
    op.p[3] = (_ptr(_h(res)) + 2 * res_off) if res is not None else None
    op.p[4] = (_ptr(_h(out)) + 2 * out_off) if out is not None else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == Nout
    if pro == 2:
        assert gn_acc_ptr is not None
        op.p[7] = int(gn_acc_ptr)
    if ntr:
        assert out_t is not None and ntr % 32 == 0
        op.p[8] = _ptr(_h(out_t))
    if order is None:
        order = int(Nout * K > M * K)                  # weight-band major per XCD when the weights outweigh the activations
    vals = [M, K, Nout, ldx, ldo, ldr, epi, pro, 0, T, G, 0, NW, NT, MT, ntr // 32, ldt, order]
    for j, v in enumerate(vals):
        op.i[j] = int(v)
    op.l[0] = int(st)
    op.f[0] = float(eps)
    return op, (x, w, bias, res, out, out_t)


# ----------------------------------------------------------------------------- weight-streaming GEMM (csrc/wsgemm.hip)
WS_BM = 128


def pack_fragments(w2d: torch.Tensor) -> torch.Tensor:
    """[Nout, K] (Nout % 32 == 0, K % 16 == 0) -> MFMA-fragment order: flat index (((t * S + s) * 64 + lane) * 8 + e) holds
    W[32 t + lane % 32][16 s + 8 (lane // 32) + e] (the A operand of v_mfma_f32_32x32x16_f16 is one contiguous 1 KB block)."""
    n, k = w2d.shape
    assert n % 32 == 0 and k % 16 == 0, (n, k)
    return w2d.to(torch.float16).view(n // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def pack_wsgemm(w: torch.Tensor, bias: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None,
                beta: Optional[torch.Tensor] = None, geglu: bool = False):
    """Linear layer [Nout, K] (K % 64 == 0, any K) for wsgemm.hip -> (fragment-packed fp16 weights, fp32 bias or None, fp32 column
    sums or None).  With `gamma` / `beta` (the LayerNorm in front of the layer) the affine map is folded in (W' = W diag(gamma),
    b' = b + W beta) and `colsum[n] = sum_k fp16(W'[n][k])` is returned for the kernel's accumulator-side normalisation
    out = rstd (x W'^T - mean colsum) + b' -- the sums are taken over the ROUNDED weights, i.e. over exactly what the matrix
    cores multiply.  `geglu`: rows re-ordered by rowgemm_geglu_perm (bias and sums likewise)."""
    w = w.reshape(w.shape[0], -1).float()
    n, k = w.shape
    assert n % 32 == 0 and k % 64 == 0, (n, k)
    b = None if bias is None else bias.float().clone()
    if gamma is not None:
        if beta is not None:
            shift = w @ beta.float()
            b = shift if b is None else b + shift
        w = w * gamma.float()[None, :]
    if geglu:
        perm = rowgemm_geglu_perm(n // 2, w.device)
        w = w[perm]
        b = None if b is None else b[perm]
    w16 = w.to(torch.float16)
    cs = w16.float().sum(1).contiguous() if gamma is not None else None
    return pack_fragments(w16), (None if b is None else b.contiguous()), cs


def pack_wsgemm_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] (Cout % 32 == 0) -> fragment-packed weights with k = tap * CinP + ci (CinP = round_up(Cin, 64))."""
    return pack_fragments(pack_conv3x3(w))


def wsgemm_ok(M: int, Nout: int, C1: int, C2: int = 0, T: int = 0) -> bool:
    """Shapes wsgemm.hip takes: 32-row weight tiles, 64-channel chunks per input, samples made of whole 32-token MFMA tiles."""
    return Nout % 32 == 0 and Nout >= 32 and C1 % 64 == 0 and C1 > 0 and C2 % 64 == 0 and (T <= 0 or T % 32 == 0) and M < (1 << 22)


def _ws_lds(NW, NT, epi, ntr, gn):
    BNp = NW * NT * 32
    BNo = BNp // 2 if epi == 1 else BNp
    ring = 4 * WS_BM * 64 * 2
    ep = WS_BM * (BNo + 8) * 2 + ((64 * NW * 32 + BNo * 4) if gn else 0)
    if ntr:
        ep = max(ep, BNp * (WS_BM + 8) * 2)
    return ((max(ring, ep) + 255) // 256) * 256 + 2 * WS_BM * 4 + 64 + 6 * 320 * 4


def wsgemm_schedule(M: int, Ktot: int, Nout: int, ntr: int = 0, epi: int = 0, pro: int = 0, taps: int = 1):
    """(NW, NT, NL, S, ntw) for a wsgemm launch: consumer waves per block, 32-row weight tiles per wave, loader waves, K slices,
    non-temporal weight loads.  The table measured in the frame (wsgemm_tuned.json, tools/wsgemm_tune.py) when it holds the
    shape, else a cost model: a block streams BN x Kc weights (HBM, ~25 B/clk/CU chip-wide 6.4 TB/s) and 128 x Kc activations
    (L2), runs NT * 4 * Kc / 16 MFMAs of 32 cycles per wave; blocks run in rounds over 256 CUs; each slice beyond the first adds
    a slab write + read of 128 x BN fp32 for the last arriver."""
    tiles = Nout // 32
    nm = (M + WS_BM - 1) // WS_BM
    nch = Ktot // 64
    force = os.environ.get("L2D_WSGEMM_FORCE")        # "NW,NT,NL,S" (tools): applied where it divides the shape
    ntw = nm == 1
    key = wsgemm_key(taps, M, Ktot, Nout, ntr, epi, pro)
    cands = []
    for nt in (1, 2):                                  # 32-row weight tiles per consumer wave (2: at most 4 consumer waves)
        for nw in range(1, 11 if nt == 1 else 5):
            if tiles % (nw * nt) or (ntr // 32) % (nw * nt):
                continue
            if _ws_lds(nw, nt, epi, ntr, True) > 163840:
                continue
            cands.append((nw, nt))
    assert cands, (M, Ktot, Nout, ntr, epi, pro)
    if force:
        nw, nt, nl, S = (int(v) for v in force.split(","))
        if (nw, nt) in cands and not (ntr and S > 1):
            return nw, nt, nl, max(1, min(S, nch)), ntw
    if key in _WS_TUNED:
        nw, nt, nl, S = _WS_TUNED[key][:4]
        if (nw, nt) in cands:
            return nw, nt, nl, max(1, min(S, nch)), ntw
    # default (shapes the table does not hold): two loader waves; about one block per CU (240-256 blocks); one weight tile per
    # wave; the channel tile grows with the length of the contraction (a long K wants few, fat slices: the activation chunks are
    # re-read by every channel tile) -- the pattern of the in-frame picks at cfg-2 (profiles/round4_a_wsgemm_tune_in_frame_cfg2.txt)
    best, best_c = None, None
    for nw, nt in cands:
        if nt != 1:
            continue
        ny = tiles // nw
        for S in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            if S > 1 and (ntr or nch // S < 4):
                continue
            blocks = nm * ny * S
            import math
            c = abs(math.log(blocks / 248.0)) + 0.12 * math.log2(S) + (0.25 if blocks > 320 else 0.0)
            want_bn = 32 if nch <= 24 else (64 if nch <= 96 else 128)       # K <= 1536: 32 channels per block, K <= 6144: 64, longer: 128
            c += 0.2 * abs(math.log2(32 * nw / want_bn))
            if best_c is None or c < best_c:
                best, best_c = (nw, 1, 2, S, ntw), c
    if best is None:
        nw, nt = cands[0]
        best = (nw, nt, 2, 1, ntw)
    return best


def _load_ws_tuned():
    import json
    path = os.environ.get("L2D_WSGEMM_TABLE") or os.path.join(os.path.dirname(__file__), "wsgemm_tuned.json")   # (override: A/B runs of tools)
    if os.environ.get("L2D_WSGEMM_NO_TABLE") or not os.path.exists(path):
        return {}, set(), set()
    with open(path) as f:
        d = json.load(f)
    return d["shapes"], set(d.get("skip", [])), set(d.get("large", []))


# shapes -> schedule; skip = few-token shapes (M <= WS_SMALL_M) where the round-3 kernel measured faster; large = shapes with MORE
# tokens where the weight-streaming kernel measured faster (there the round-3 kernels are the default: opt-in, not opt-out)
_WS_TUNED, _WS_SKIP, _WS_LARGE = _load_ws_tuned()
_WS_TUNED_M = {int(k.split(",")[1]) for k in list(_WS_TUNED) + list(_WS_SKIP)}      # token counts the tuner measured (any shape class)
WS_SMALL_M = 1280


def wsgemm_key(taps: int, M: int, Ktot: int, Nout: int, ntr: int = 0, epi: int = 0, pro: int = 0) -> str:
    return f"{taps},{M},{Ktot},{Nout},{ntr},{epi},{pro}"


def wsgemm_wanted(taps: int, M: int, Ktot: int, Nout: int, ntr: int = 0, epi: int = 0, pro: int = 0) -> bool:
    """Few-token shapes (M <= 1280): True unless the in-frame tuner measured the round-3 kernel (igemm / rowgemm) faster than the best
    wsgemm schedule (`skip` list of wsgemm_tuned.json, tools/wsgemm_tune.py).  Shapes with more tokens (round 5: level 1 of the
    BASELINE configs, 2048-4608 tokens): True only where the tuner measured the weight-streaming kernel faster (`large` list) --
    an untuned resolution keeps the round-3 kernels there.  The packer follows this."""
    key = wsgemm_key(taps, M, Ktot, Nout, ntr, epi, pro)
    # measured by the tuner?  few tokens: every candidate shape it saw is in `shapes` or `skip`; above 1280 tokens it records winners
    # only, so there the token count stands for "this level was offered"
    tuned = (key in _WS_TUNED or key in _WS_SKIP) if M <= WS_SMALL_M else (M in _WS_TUNED_M)
    if not tuned and os.environ.get("L2D_WSGEMM_RULE", "1") != "0" and not os.environ.get("L2D_WSGEMM_LARGE_ALL"):
        return _wsgemm_wanted_rule(taps, M, Ktot, Nout, ntr, epi, pro)
    if M > WS_SMALL_M and not os.environ.get("L2D_WSGEMM_LARGE_ALL"):      # (LARGE_ALL: the tuner offers every shape and measures)
        return key in _WS_LARGE
    return key not in _WS_SKIP


def _wsgemm_wanted_rule(taps: int, M: int, Ktot: int, Nout: int, ntr: int, epi: int, pro: int) -> bool:
    """Token counts the in-frame tuner never saw (any resolution outside the five BASELINE configs): the pattern of its picks there
    (122 of the 150 candidate shapes of the five plans follow it; the rest are within its 3 % threshold either way).  More than 1280
    tokens (level 1): LayerNorm + q|k|v and LayerNorm + GEGLU, the long plain contractions from 3072 tokens on, the long 3x3 convs
    that cconv.hip does not take.  Fewer: everything at the 1280-wide levels; at 640 wide the LayerNorm + q|k|v / GEGLU layers from
    512 tokens on and the 640 -> 1280 shortcuts; nothing at 320 wide (cfg-1: the row GEMM wins every layer of its levels 0 / 1).
    L2D_WSGEMM_RULE=0: the pre-round-6 default (every few-token shape, nothing above 1280 tokens; profiles/round6_w_*)."""
    K = Ktot // taps
    if M > 4608:                # (beyond every token count the kernel was measured at; the plan's own bound is L2D_WSGEMM_MAX_M)
        return False
    if M > WS_SMALL_M:
        if taps == 9:
            return Ktot >= 5760 or M >= 3072
        if pro == 1:
            return K == 640 and Nout >= 1920
        return epi == 0 and Ktot >= 960 and Nout >= 640 and M >= 3072       # (the 320-wide level was never measured on this kernel)
    if taps == 9:
        return True
    if K <= 320:
        return False
    if K <= 640:
        return (pro == 1 and Nout >= 1920 and M >= 512) or (pro == 0 and epi == 0 and Nout >= 2 * K)
    return True


def wsgemm_sizes(M: int, Nout: int, NW: int, NT: int, S: int):
    """(fp32 workspace elements, int32 counters) of a split-K wsgemm launch: one slab of 128 x BN partial sums + 256 statistics
    floats per (channel tile, row tile, slice)."""
    nm = (M + WS_BM - 1) // WS_BM
    ny = (Nout // 32) // (NW * NT)
    return nm * ny * S * (WS_BM * NW * NT * 32 + 2 * WS_BM), nm * ny


def wsgemm(x1, w, out, *, M, Nout, C1, ldx1, ldo, x2=None, C2=0, ldx2=0, bias=None, colsum=None, rowbias=None, ldrb=0,
           rows_per_bias=0, res=None, ldr=0, taps=1, B=1, H=1, W=1, epi=0, pro=0, eps=1e-5, T=0, out_t=None, ntr=0, ldt=0,
           st=0, sched=None, ws=None, cnt=None, cnt_off=0, x1_off=0, out_off=0, res_off=0):
    """Weight-streaming GEMM (csrc/wsgemm.hip): out[m][n] = epi(sum_k LN?(x)[m][k] W[n][k]) for M <~ 1k tokens; linear layers
    (taps = 1, optional two-input channel concat, LayerNorm fold pro = 1 with `colsum`, GEGLU epi = 1, trailing `ntr` packed rows
    stored transposed into out_t) and 3x3 stride-1 pad-1 convs (taps = 9, x = [B, H, W, C1 (+ C2)], per-sample `rowbias`).
    `w` / `bias` / `colsum` come from pack_wsgemm / pack_wsgemm_conv3x3.  sched = (NW, NT, NL, S, ntw) or None for
    wsgemm_schedule; S > 1 needs `ws` (wsgemm_sizes()[0] floats) and `cnt` (int32 counters, zero, wsgemm_sizes()[1] from cnt_off)."""
    op = L2dOp()
    op.kind = _lib.OP_WSGEMM
    CinP = C1 + C2
    Ktot = taps * CinP
    assert w.dtype == torch.float16 and w.numel() == Nout * Ktot, (w.shape, Nout, Ktot)
    if sched is None:
        sched = wsgemm_schedule(M, Ktot, Nout, ntr, epi, pro, taps)
    NW, NT, NL, S, ntw = sched
    zp = zero_page(x1.device)
    op.p[0] = _ptr(_h(x1)) + 2 * x1_off
    op.p[1] = _ptr(x2) if x2 is not None else None
    op.p[2] = _ptr(_h(w))
    op.p[3], op.p[4] = _ptr(bias), _ptr(rowbias)
    op.p[5] = (_ptr(_h(res)) + 2 * res_off) if res is not None else None
    op.p[6] = (_ptr(_h(out)) + 2 * out_off) if out is not None else None
    op.p[7] = _ptr(zp)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == Nout
    if pro == 1:
        assert colsum is not None and colsum.dtype == torch.float32 and colsum.numel() == Nout
        op.p[13] = _ptr(colsum)
    if ntr:
        assert out_t is not None and ntr % 32 == 0
        op.p[8] = _ptr(_h(out_t))
    if S > 1:
        need_ws, need_cnt = wsgemm_sizes(M, Nout, NW, NT, S)
        assert ws is not None and ws.dtype == torch.float32 and ws.numel() >= need_ws, (need_ws,)
        assert cnt is not None and cnt.dtype == torch.int32 and cnt.numel() >= cnt_off + need_cnt
        op.p[11] = _ptr(cnt) + 4 * cnt_off
        op.p[12] = _ptr(ws)
    vals = {0: taps, 1: C1, 2: C2, 3: ldx1, 4: ldx2, 5: CinP, 6: B, 7: H, 8: W, 9: NW, 10: NT, 11: NL, 12: S, 13: M, 14: Nout,
            15: ldo, 16: ldr, 17: ldrb, 18: rows_per_bias, 19: epi, 20: pro, 21: ntr // 32, 22: ldt, 23: int(bool(ntw)), 30: T}
    for j, v in vals.items():
        op.i[j] = int(v)
    op.l[0] = int(st)
    op.f[0] = float(eps)
    return op, (x1, x2, w, bias, colsum, rowbias, res, out, out_t, zp, ws, cnt)


def wsgemm_gn_target(op, acc_ptr: int, *, T: int, G: int, cpg: int, choff: int) -> bool:
    """GroupNorm statistics of a wsgemm launch's output for a consumer GroupNorm (a 128-token tile may span samples: T % 32)."""
    assert op.kind == _lib.OP_WSGEMM
    NW, NT, ntr, epi, M = op.i[9], op.i[10], op.i[21], op.i[19], op.i[13]
    bno = NW * NT * 32
    if T % 32 or M % T or ntr or epi == 1 or G > 32 or (cpg | choff) & 1 or 64 * NW < bno // 2:
        return False
    if op.p[9] and (op.i[24], op.i[25]) != (T, G):
        return False
    slot = 0 if not op.p[9] else (1 if not op.p[10] else -1)
    if slot < 0:
        return False
    op.p[9 + slot] = int(acc_ptr)
    op.i[24], op.i[25] = int(T), int(G)
    op.i[26 + 2 * slot], op.i[27 + 2 * slot] = int(cpg), int(choff)
    return True


# ----------------------------------------------------------------------------- cconv (csrc/cconv.hip, round 6)
CCONV_RING = 9          # weight ring depth of the kernel in k steps: the packed tensor is padded by this many 2 KB steps


def pack_cconv(w: torch.Tensor, KG: int) -> torch.Tensor:
    """[Cout, Cin, 3, 3] (Cout % 64 == 0; Cin padded to a multiple of 64) -> the weight STREAMS of cconv.hip: for every
    (64-channel tile n64, K group kg) one contiguous sequence [chunk c][tap t][u < 4 / KG][half i < 2][64 lanes][8 halfs] holding
    W[64 n64 + 32 i + lane % 32][tap t][64 c + 16 (u KG + kg) + 8 (lane // 32) + e] -- two A operands of
    v_mfma_f32_32x32x16_f16 per k step, in exactly the order a compute wave consumes them.  Padded with CCONV_RING k steps of
    zeros (the ring of a block's last k steps requests beyond its slice)."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cout % 64 == 0 and KG in (1, 2, 4)
    cinp = round_up(cin, 64)
    wp = torch.zeros(cout, 9, cinp, dtype=torch.float16, device=w.device)
    wp[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin).to(torch.float16)
    upt = 4 // KG
    # [n64][i][r32][t][c][u][kg][lh][e] -> [n64][kg][c][t][u][i][lh][r32][e]
    v = wp.view(cout // 64, 2, 32, 9, cinp // 64, upt, KG, 2, 8).permute(0, 6, 4, 3, 5, 1, 7, 2, 8).contiguous().view(-1)
    return torch.cat([v, torch.zeros(CCONV_RING * 1024, dtype=torch.float16, device=w.device)])


def unpack_cconv(packed: torch.Tensor, cout: int, cinp: int, KG: int) -> torch.Tensor:
    """inverse of pack_cconv (tests): -> [Cout, 9, CinP]"""
    upt = 4 // KG
    v = packed[:cout * 9 * cinp].view(cout // 64, KG, cinp // 64, 9, upt, 2, 2, 32, 8)
    return v.permute(0, 5, 7, 3, 2, 4, 1, 6, 8).reshape(cout, 9, cinp)


def cconv_ok(H: int, W: int, Nout: int, C1: int, C2: int = 0) -> bool:
    """Shapes cconv.hip takes: whole 8 x 16 patches of OUTPUT pixels, 64-channel chunks per input, 64-channel output tiles."""
    return H % 8 == 0 and W % 16 == 0 and Nout % 64 == 0 and C1 % 64 == 0 and C1 > 0 and C2 % 64 == 0


def cconv_schedule(B: int, H: int, W: int, Nout: int, CinP: int, KG: Optional[int] = None):
    """(CG, KG, NLD, S) for a cconv launch: channel tile (64 CG), K groups per tile (CG KG = 4 compute waves), loader waves and
    K slices of whole 64-channel chunks, from a cost model fitted to tools/cconv_time.py / tools/cconv_stamps.py (MI355X): the k loop
    of the longest slice at ~0.7 of the matrix rate, a fixed prologue / epilogue, per split the slab publish and the last arriver's
    sum, and a second round of blocks beyond one per CU.  `KG`: the packing's K-group count when it is already fixed (the warm-up
    plan shares the stream plan's packed weights).  L2D_CCONV_FORCE="CG,KG,NLD,S" overrides (tuning)."""
    force = os.environ.get("L2D_CCONV_FORCE")
    nch = CinP // 64
    if force:
        cg, kg, nld, S = (int(v) for v in force.split(","))
        if (KG is None or kg == KG) and Nout % (64 * cg) == 0:
            return cg, kg, nld, max(1, min(S, nch))
    npat = B * (H // 8) * (W // 16)
    best = None
    for cg, kg, nld in ((1, 4, 4), (2, 2, 4)):
        if Nout % (64 * cg) or (KG is not None and kg != KG):
            continue
        tiles = npat * (Nout // (64 * cg))
        for S in range(1, min(nch, 8) + 1):
            blocks = tiles * S
            if blocks > 256 and S > 1:
                break
            # shader cycles (tools/cconv_stamps.py): a chunk is 3.1 k (KG = 4: 72 MFMAs per wave) / 6.2 k (KG = 2: 144) cycles, a block
            # pays ~12 k for prologue + K-group reduction + epilogue, a split adds the ticket / publish overlap and one slab read per
            # other slice; blocks beyond one per CU run as a second round
            chunks = -(-nch // S)
            loop = chunks * (3100 if kg == 4 else 6200)
            tail = ((10000 if cg == 2 else 9000) + (S - 1) * (4500 if cg == 2 else 2500)) if S > 1 else 0
            rounds = -(-blocks // 256)
            cost = rounds * (loop + 12000) + tail
            if best is None or cost < best[0]:
                best = (cost, (cg, kg, nld, S))
    assert best is not None, (B, H, W, Nout, CinP, KG)
    return best[1]


def cconv_wanted(B: int, H: int, W: int, Cin: int, Nout: int, ups: int = 0) -> bool:
    """Whether the plan gives a 3x3 stride-1 conv to cconv.hip (H x W = output resolution).  Measured at cfg-2 against the kernel
    each launch had before (tools/cconv_time.py, profiles/round6_a_cconv_time.log): the up-samplers 1.4-1.5 x (igemm), the resnet
    convs of the 640- / 1280-wide levels 1.2-1.6 x (wsgemm); the 320-wide level keeps the patch conv by default (1.02-1.14 x in isolation,
    nothing in the frame: 320 blocks of one 64-channel tile are two rounds over the chip; L2D_CCONV_L0=1 moves it too).  L2D_CCONV=0
    switches the kernel off (A/B)."""
    if os.environ.get("L2D_CCONV", "1") == "0" or not cconv_ok(H, W, Nout, Cin):
        return False
    if ups:
        return True
    return B * H * W >= 512 and (Nout >= 640 or os.environ.get("L2D_CCONV_L0", "0") != "0")


def cconv_sizes(B: int, H: int, W: int, Nout: int, CG: int, S: int):
    """(fp32 workspace elements, int32 counters) of a split-K cconv launch: one 128 x 64 CG slab per (tile, slice), a ticket and a
    done counter per tile"""
    tiles = B * (H // 8) * (W // 16) * (Nout // (64 * CG))
    return tiles * S * 128 * 64 * CG, 2 * tiles


def cconv(x1, w, out, *, B, H, W, C1, ldx1, Nout, ldo, KG, x2=None, C2=0, ldx2=0, ups=0, bias=None, rowbias=None, ldrb=0,
          rows_per_bias=0, res=None, ldr=0, sched=None, ws=None, cnt=None, cnt_off=0, gn_acc_ptr=None, gn_gamma=None, gn_beta=None,
          gn_G=0, gn_eps=1e-5):
    """3x3 stride-1 pad-1 conv with the activation patch resident in LDS and register-streamed weights (csrc/cconv.hip).
    H x W = OUTPUT resolution; ups = 1: the input is [B, H/2, W/2, C] and is up-sampled x2 (nearest) on the fly (Upsample3D).
    `w` = pack_cconv(weight, KG).  sched = (CG, KG, NLD, S) or None for cconv_schedule (its KG must equal the packing's);
    S > 1 needs `ws` / `cnt` (cconv_sizes).  gn_acc_ptr (+ gn_gamma / gn_beta fp16 [C1 + C2], gn_G, gn_eps): the conv of
    silu(GroupNorm(x)) -- the normalisation runs in the kernel's loader waves from the producers' fixed-point statistics
    (int64 [B][G][2], as gn_apply with nchunk = 0): no GroupNorm launch, no normalised tensor in HBM."""
    op = L2dOp()
    op.kind = _lib.OP_CCONV
    CinP = C1 + C2
    if sched is None:
        sched = cconv_schedule(B, H, W, Nout, CinP)
    CG, KG_, NLD, S = sched
    assert KG_ == KG, (sched, KG)
    assert w.dtype == torch.float16 and w.numel() == Nout * 9 * CinP + CCONV_RING * 1024, (w.shape, Nout, CinP)
    zp = zero_page(x1.device)
    op.p[0], op.p[1], op.p[2] = _ptr(_h(x1)), (_ptr(x2) if x2 is not None else None), _ptr(_h(w))
    op.p[3], op.p[4], op.p[5] = _ptr(bias), _ptr(rowbias), (_ptr(_h(res)) if res is not None else None)
    op.p[6], op.p[7] = _ptr(_h(out)), _ptr(zp)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == Nout
    if S > 1:
        need_ws, need_cnt = cconv_sizes(B, H, W, Nout, CG, S)
        assert ws is not None and ws.dtype == torch.float32 and ws.numel() >= need_ws, (need_ws,)
        assert cnt is not None and cnt.dtype == torch.int32 and cnt.numel() >= cnt_off + need_cnt
        op.p[11] = _ptr(cnt) + 4 * cnt_off
        op.p[12] = _ptr(ws)
    vals = {1: C1, 2: C2, 3: ldx1, 4: ldx2, 5: CinP, 6: B, 7: H, 8: W, 9: CG, 10: KG, 11: NLD, 12: S, 13: ups, 14: Nout, 15: ldo,
            16: ldr, 17: ldrb, 18: rows_per_bias}
    if gn_acc_ptr is not None:
        assert gn_gamma.dtype == torch.float16 and gn_beta.dtype == torch.float16 and gn_gamma.numel() == CinP and gn_beta.numel() == CinP
        op.p[13], op.p[14], op.p[15] = int(gn_acc_ptr), _ptr(gn_gamma), _ptr(gn_beta)
        vals.update({20: 1, 21: gn_G})
        op.f[0] = float(gn_eps)
    for j, v in vals.items():
        op.i[j] = int(v)
    return op, (x1, x2, w, bias, rowbias, res, out, zp, ws, cnt, gn_gamma, gn_beta)


def cconv_gn_target(op, acc_ptr: int, *, T: int, G: int, cpg: int, choff: int) -> bool:
    """GroupNorm statistics of a cconv launch's output for a consumer GroupNorm (a patch never straddles samples)."""
    assert op.kind == _lib.OP_CCONV
    if T != op.i[7] * op.i[8] or G > 32 or (cpg | choff) & 1:
        return False
    if op.p[9] and (op.i[24], op.i[25]) != (T, G):
        return False
    slot = 0 if not op.p[9] else (1 if not op.p[10] else -1)
    if slot < 0:
        return False
    op.p[9 + slot] = int(acc_ptr)
    op.i[24], op.i[25] = int(T), int(G)
    op.i[26 + 2 * slot], op.i[27 + 2 * slot] = int(cpg), int(choff)
    return True


def igemm_schedule(M: int, Nout: int, Kp: int, batch: int = 1, epi: int = 0, taps: int = 1):
    """(tile, splitk, variant) for an implicit GEMM, from the on-GPU sweep (tools/igemm_sweep.py, profiles/r1b_igemm_sweep.txt).
    tile 1 = 128x128, 2 = 64x64; variant = pipeline shape (igemm.hip launch_p).  These kernels are occupancy /
    latency bound, not DMA-depth bound: what matters is >= ~1.5 waves of blocks over the 256 CUs.
      * >= 384 big tiles: 128x128 (3 blocks/CU with the 48 KB BK32x3 ring when there are >= 768 tiles);
      * otherwise 64x64 tiles if that yields >= 256 blocks;
      * otherwise (low-resolution levels: M = 128..2048 tokens, K up to 23 040) split K over 128x128 tiles
        (>= 8 BK64 steps per split, K >= 2048), reduced by igemm_splitk_epilogue."""
    key = f"{taps},{M},{Nout},{Kp},{epi},{batch}"
    force = os.environ.get("L2D_IGEMM_FORCE")       # "tile,S,variant": exploration runs of tools/igemm_pick.py
    if force:
        t, S, v = (int(x) for x in force.split(","))
        S = max(1, min(S, Kp // 128))
        ok = not (v in (6, 7) and (Kp // taps) % 128) and not (t == 1 and v in (7, 8, 9)) and 0 <= v <= 10
        if ok:
            return t, S, v
    elif key in _TUNED:
        return tuple(_TUNED[key])
    if os.environ.get("L2D_IGEMM_HEUR", "6") == "1":
        return _igemm_heuristic_r1(M, Nout, Kp, batch, epi)
    return _igemm_heuristic_r6(M, Nout, Kp, batch, epi)


def _igemm_heuristic_r6(M: int, Nout: int, Kp: int, batch: int, epi: int):
    """Fallback schedule for shapes the tuned table does not hold (every resolution outside cfg-2), round 6: the rule that the 60
    in-frame picks of igemm_tuned.json follow, instead of the round-1 sweep's.  What the table says: few-token launches want
    64 x 64 tiles with K split until ~240 blocks exist (~480 for K >= 2560) but never below ~6 BK64 steps per split nor above 6
    splits unless a split would still be longer than ~30 steps; 128 x 128 tiles + split-K only for the long 3x3 contractions
    (K >= 8640) at >= 512 tokens; no split for GEGLU epilogues.  L2D_IGEMM_HEUR=1 restores the round-1 rule (A/B:
    profiles/round6_u_*)."""
    cdiv = lambda a, b: (a + b - 1) // b
    nk64 = Kp // 64
    t64 = cdiv(Nout, 64) * cdiv(M, 64) * batch
    t128 = cdiv(Nout, 128) * cdiv(M, 128) * batch
    if Nout <= 64 and M >= 16384:
        return 2, 1, (1 if nk64 <= 18 else 2)
    if t128 >= 384:
        return 1, 1, (4 if t128 >= 768 else 5)
    if epi == 1:
        return 2, 1, 1
    if Kp >= 8640 and M >= 512 and M * Kp >= 8_000_000:
        return 1, max(1, min(480 // t128, nk64 // 8, 16)), 5
    if t64 >= 256:
        return 2, (max(1, min(nk64 // 24, 1280 // t64)) if nk64 >= 64 else 1), 1
    target = 240 if Kp <= 1920 else 480
    return 2, max(1, min(target // t64, max(6, cdiv(nk64, 30)), nk64 // 6, 16)), 1


def _igemm_heuristic_r1(M: int, Nout: int, Kp: int, batch: int, epi: int):
    """The round-1 rule (tools/igemm_sweep.py, isolated launches): kept for A/B."""
    v_small, v_big = 1, 5           # BK64 x 3 stages: in-frame best (85.3 vs 79.9 fps with x2); 128x128 BK32 x 4 for the big shapes
    cdiv = lambda a, b: (a + b - 1) // b
    nk64 = Kp // 64
    big = cdiv(Nout, 128) * cdiv(M, 128) * batch
    if Nout <= 64 and M >= 16384:
        # narrow outputs (TAESD: 64 channels everywhere, 3 / 4 at the ends): a 128-wide channel tile would be half empty
        return 2, 1, (v_small if nk64 <= 18 else 2)
    if big >= 384:
        return 1, 1, (4 if big >= 768 else v_big)
    if epi == 1:
        return 2, 1, v_small
    if nk64 >= 32:
        # long K (3x3 convs below the top level, FF down-projections): 128x128 tiles + split-K beat 64x64 tiles
        # (level-1 conv in the frame: 77 us with 320 small tiles vs 45 us with 80 big tiles x 3 splits)
        s_big = max(1, min(nk64 // 8, round(256 / big), 64))
        if s_big >= 2:
            return 1, s_big, v_big
    return 2, 1, v_small


def _load_tuned():
    """Per-shape (tile, split-K, pipeline variant) picks measured IN-FRAME on MI355X (tools/igemm_pick.py over
    rocprofv3 traces of whole frames): the heuristic above is the fallback for shapes the table does not hold."""
    import json
    path = os.path.join(os.path.dirname(__file__), "igemm_tuned.json")
    if os.environ.get("L2D_IGEMM_NO_TABLE") or not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)["shapes"]


_TUNED = _load_tuned()


def igemm(x1, w, out, *, M, Nout, C1, ldx1, CinP, ldo, x2=None, C2=0, ldx2=0, bias=None, rowbias=None, ldrb=0,
          rows_per_bias=0, res=None, ldr=0, taps=1, B=1, Hin=1, Win=1, Hout=1, Wout=1, stride=1, ups=0, epi=0,
          batch=1, sx1=0, sw=0, so=0, sres=0, x1_off=0, w_off=0, out_off=0, res_off=0, splitk=1, tile=0, ws=None,
          variant=5, order=0, pad_same=False, cnt=None, cnt_off=0):
    """Offsets (in elements) allow sub-views of fp16 buffers without creating tensors.
    splitk > 1 needs `ws`: fp32 workspace of batch * splitk * M * round_up(Nout, 4) elements (two-launch reduction), or, with
    `cnt` (int32 arrival counters, zero; this launch uses splitk_sizes(...)[1] of them from cnt_off), the fused reduction's
    splitk_sizes(...)[0] elements: the last-arriving block of a tile reduces and runs the epilogue, no second launch."""
    op = L2dOp()
    op.kind = _lib.OP_IGEMM
    es = 2
    zp = zero_page(x1.device)
    op.p[0] = _ptr(_h(x1)) + x1_off * es
    op.p[1] = _ptr(x2) if x2 is not None else None
    op.p[2] = _ptr(_h(w)) + w_off * es
    op.p[3] = _ptr(bias)
    op.p[4] = _ptr(rowbias)
    op.p[5] = (_ptr(res) + res_off * es) if res is not None else None
    op.p[6] = _ptr(_h(out)) + out_off * es
    op.p[7] = _ptr(zp)
    op.p[8] = _ptr(ws)
    if bias is not None:
        assert bias.dtype == torch.float32
    if rowbias is not None:
        assert rowbias.dtype == torch.float32
    if splitk > 1 and cnt is not None:
        need_ws, need_cnt = splitk_sizes(M, Nout, splitk, batch, tile)
        assert tile in (1, 2) and ws is not None and ws.dtype == torch.float32 and ws.numel() >= need_ws
        assert cnt.dtype == torch.int32 and cnt.numel() >= cnt_off + need_cnt
        op.p[11] = _ptr(cnt) + 4 * cnt_off
    elif splitk > 1:
        assert ws is not None and ws.dtype == torch.float32 and ws.numel() >= batch * splitk * M * round_up(Nout, 4)
    direct_epi = 0                  # (bit 5 of i22 selects the round-1 register -> global epilogue: A/B settled in round 2)
    vals = [taps, C1, C2, ldx1, ldx2, CinP, B, Hin, Win, Hout, Wout, stride, ups, M, Nout, ldo, ldr, ldrb,
            rows_per_bias, epi, batch, splitk, int(tile) + 16 * int(order) + direct_epi, variant]
    for j, v in enumerate(vals):
        op.i[j] = int(v)
    op.l[0], op.l[1], op.l[2], op.l[3] = int(sx1), int(sw), int(so), int(sres)
    op.i[30] = 1 if pad_same else 0       # TF-"SAME" low-side padding 0 (stride-2 3x3 convs of the ResNetV2 backbone)
    return op, (x1, x2, w, bias, rowbias, res, out, zp, ws, cnt)


def splitk_sizes(M: int, Nout: int, splitk: int, batch: int, tile: int):
    """(fp32 workspace elements, int32 counters) of a split-K igemm with the fused reduction: whole tiles, one slab per split."""
    t = 128 if tile == 1 else 64
    ntiles = ((M + t - 1) // t) * ((Nout + t - 1) // t)
    return batch * ntiles * splitk * t * t, batch * ntiles


SPLITK_FUSED = True              # (False = separate reduction launch, the round-1 form: A/B settled in round 3)
SPLITK_FUSED_MAX = 16            # deeper splits keep the reduction launch: ONE block
# per tile sums all S slabs in the fused form, which serialises when a launch has few tiles and many splits (8x8 levels)


def splitk_fused(splitk: int) -> bool:
    return SPLITK_FUSED and 1 < splitk <= SPLITK_FUSED_MAX


def gn_stats(x1, partial, *, B, T, C1, ld1, G, nchunk, x2=None, C2=0, ld2=0):
    op = L2dOp()
    op.kind = _lib.OP_GN_STATS
    op.p[0], op.p[1], op.p[2] = _ptr(_h(x1)), _ptr(x2), _ptr(partial)
    for j, v in enumerate([B, T, C1, C2, ld1, ld2, G, nchunk, 0]):
        op.i[j] = int(v)
    return op, (x1, x2, partial)


ACT_NONE, ACT_SILU, ACT_RELU, ACT_ADD_RELU = 0, 1, 2, 3


def gn_self_ok(T: int, C: int, G: int) -> bool:
    """Shapes the ONE-launch GroupNorm takes (norm.hip gn_self_kernel: gn_apply with nchunk = 0 and no accumulator): a block holds all T
    pixel rows of a band of whole groups (lcm(cpg, 8) <= 128 channels, <= 4 groups) in registers, <= 16 vectors per thread.
    L2D_GN_SELF=0 keeps the two-launch fallback (A/B)."""
    if os.environ.get("L2D_GN_SELF", "1") == "0" or G <= 0 or C % G:
        return False
    cpg = C // G
    band = cpg
    while band % 8:
        band += cpg
    if band > 128 or band // cpg > 4 or C % band:
        return False
    pr = 256 // (band // 8)
    return -(-T // pr) <= 16


def gn_apply(x1, partial, gamma, beta, out, *, B, T, C1, ld1, G, nchunk, eps, silu, x2=None, C2=0, ld2=0, acc_ptr=None, res=None):
    """nchunk = 0 + acc_ptr: the statistics come from the fixed-point accumulators [B][G][2] int64 that the producing igemm
    launches filled (igemm `gn_target`), `partial` is unused (None).  nchunk = 0 WITHOUT acc_ptr (and partial None): the one-launch
    form for small tensors -- statistics inside the launch (gn_self_ok)."""
    op = L2dOp()
    op.kind = _lib.OP_GN_APPLY
    op.p[0], op.p[1], op.p[2] = _ptr(_h(x1)), _ptr(x2), _ptr(partial)
    if acc_ptr is not None:
        assert nchunk == 0
        op.p[6] = int(acc_ptr)
    op.p[3], op.p[4], op.p[5] = _ptr(_h(gamma)), _ptr(_h(beta)), _ptr(_h(out))
    for j, v in enumerate([B, T, C1, C2, ld1, ld2, G, nchunk, int(silu)]):     # silu: bool, or an ACT_* code
        op.i[j] = int(v)
    op.f[0] = float(eps)
    if res is not None:
        op.p[7] = _ptr(_h(res))
    return op, (x1, x2, partial, gamma, beta, out, res)


def layernorm(x, gamma, beta, out, *, rows, C, ldx, ldo, eps=1e-5):
    op = L2dOp()
    op.kind = _lib.OP_LAYERNORM
    op.p[0], op.p[1], op.p[2], op.p[3] = _ptr(_h(x)), _ptr(_h(gamma)), _ptr(_h(beta)), _ptr(_h(out))
    for j, v in enumerate([rows, C, ldx, ldo]):
        op.i[j] = int(v)
    op.f[0] = float(eps)
    return op, (x, gamma, beta, out)


def flash_attn(q, k, vt, out, *, B, H, d, Tq, Tk, ldq, ldk, ldvt, ldo, sq, sk, svt, so, q_off=0, k_off=0, vt_off=0,
               variant=None):
    """variant: 0 auto (LDS-DMA ring kernel), 1 register-staged kernel (round 1), 2 / 3 ring with 32 / 16 query rows per
    wave, 4 ring with 32 rows per wave and the software-pipelined loop (d <= 48; other head sizes run variant 2);
    None = L2D_FLASH_VARIANT from the environment (A/B knob), default 0."""
    op = L2dOp()
    op.kind = _lib.OP_FLASH_ATTN
    zp = zero_page(q.device)
    op.p[0] = _ptr(_h(q)) + q_off * 2
    op.p[1] = _ptr(_h(k)) + k_off * 2
    op.p[2] = _ptr(_h(vt)) + vt_off * 2
    op.p[3] = _ptr(_h(out))
    op.p[4] = _ptr(zp)
    if variant is None:
        variant = int(os.environ.get("L2D_FLASH_VARIANT", "0"))
    for j, v in enumerate([B, H, d, Tq, Tk, ldq, ldk, ldvt, ldo, variant]):
        op.i[j] = int(v)
    op.l[0], op.l[1], op.l[2], op.l[3] = int(sq), int(sk), int(svt), int(so)
    return op, (q, k, vt, out, zp)


def tattn_stream(qkv, cache, q_pe, k_pe, v_pe, pe_idx, update_idx, bias, out, *, N, T, C, L, H, variant=0):
    assert cache.dtype == torch.float16 and cache.is_contiguous() and tuple(cache.shape) == (N, 2, T, L, C), \
        (cache.dtype, tuple(cache.shape), (N, 2, T, L, C))
    assert pe_idx.dtype == torch.int64 and update_idx.dtype == torch.int64 and bias.dtype == torch.float16
    op = L2dOp()
    op.kind = _lib.OP_TATTN_STREAM
    zp = zero_page(qkv.device)                      # DMA source of masked slots (ring kernel)
    for j, t in enumerate([qkv, cache, q_pe, k_pe, v_pe, pe_idx, update_idx, bias, out, zp]):
        op.p[j] = _ptr(t)
    for j, v in enumerate([N, T, C, L, H, variant]):
        op.i[j] = int(v)
    return op, (qkv, cache, q_pe, k_pe, v_pe, pe_idx, update_idx, bias, out, zp)


def tattn_warmup(qkv, cache_row, q_pe, k_pe, v_pe, out, *, F, T, C, L, H):
    assert cache_row.dtype == torch.float16 and cache_row.is_contiguous() and tuple(cache_row.shape) == (2, T, L, C)
    op = L2dOp()
    op.kind = _lib.OP_TATTN_WARMUP
    for j, t in enumerate([qkv, cache_row, q_pe, k_pe, v_pe]):
        op.p[j] = _ptr(t)
    op.p[8] = _ptr(out)
    for j, v in enumerate([F, T, C, L, H]):
        op.i[j] = int(v)
    return op, (qkv, cache_row, q_pe, k_pe, v_pe, out)


def skinny_linear(a, w, bias, out, *, M, K, Nout, silu_out=False, ldo=None, lda=0, a_off=0):
    op = L2dOp()
    op.kind = _lib.OP_SKINNY_LINEAR
    op.p[0], op.p[1], op.p[2], op.p[3] = _ptr(_h(a)) + 2 * a_off, _ptr(_h(w)), _ptr(bias), _ptr(out)
    op.l[0] = int(lda)
    is_f = out.dtype == torch.float32
    for j, v in enumerate([M, K, Nout, 1 if silu_out else 0, 1 if is_f else 0, ldo if ldo is not None else Nout]):
        op.i[j] = int(v)
    return op, (a, w, bias, out)


def timestep_embed(t, out, *, N, dim):
    assert t.dtype == torch.int64
    op = L2dOp()
    op.kind = _lib.OP_TIMESTEP_EMBED
    op.p[0], op.p[1] = _ptr(t), _ptr(_h(out))
    op.i[0], op.i[1] = N, dim
    return op, (t, out)


MAP_COPY, MAP_ADD_SCALE, MAP_TANH3, MAP_SCALE_ADD = 0, 1, 2, 3     # element maps of the layout kernels (include/l2d.h)


def nchw_to_nhwc(x, out, *, B, C, HW, Cpad, mode=MAP_COPY, a=1.0, b=0.0):
    op = L2dOp()
    op.kind = _lib.OP_NCHW_TO_NHWC
    op.p[0], op.p[1] = _ptr(_h(x)), _ptr(_h(out))
    for j, v in enumerate([B, C, HW, Cpad, mode]):
        op.i[j] = int(v)
    op.f[0], op.f[1] = float(a), float(b)
    return op, (x, out)


def nhwc_to_nchw(x, out, *, B, C, HW, ld, mode=MAP_COPY, a=1.0, b=0.0):
    op = L2dOp()
    op.kind = _lib.OP_NHWC_TO_NCHW
    op.p[0], op.p[1] = _ptr(_h(x)), _ptr(_h(out))
    for j, v in enumerate([B, C, HW, ld, mode]):
        op.i[j] = int(v)
    op.f[0], op.f[1] = float(a), float(b)
    return op, (x, out)


def stem7x7(img, w, out, *, B, H, W):
    """weight-standardised 7x7 stride-2 SAME conv 3 -> 64 from the NCHW image to channels-last (DPT-Hybrid stem)."""
    op = L2dOp()
    op.kind = _lib.OP_STEM7X7
    op.p[0], op.p[1], op.p[2] = _ptr(_h(img)), _ptr(_h(w)), _ptr(_h(out))
    op.i[0], op.i[1], op.i[2] = int(B), int(H), int(W)
    return op, (img, w, out)


RS_MAXPOOL, RS_SUBSAMPLE, RS_UP2X = 0, 1, 2


def resample_nhwc(x, out, *, B, H, W, C, mode):
    op = L2dOp()
    op.kind = _lib.OP_RESAMPLE_NHWC
    op.p[0], op.p[1] = _ptr(_h(x)), _ptr(_h(out))
    for j, v in enumerate([B, H, W, C, mode]):
        op.i[j] = int(v)
    return op, (x, out)


def ew(a, b, out, out_relu, *, n):
    """s = a (+ b); out = s; out_relu = relu(s) (either output may be None)."""
    op = L2dOp()
    op.kind = _lib.OP_EW
    op.p[0], op.p[1], op.p[2], op.p[3] = _ptr(_h(a)), _ptr(b), _ptr(out), _ptr(out_relu)
    op.l[0] = int(n)
    return op, (a, b, out, out_relu)


def resize_bilinear(x, out, *, planes, Hin, Win, Hout, Wout):
    """F.interpolate(x, (Hout, Wout), mode="bilinear", align_corners=False) on [planes, Hin, Win] fp16."""
    op = L2dOp()
    op.kind = _lib.OP_RESIZE_BILINEAR
    op.p[0], op.p[1] = _ptr(_h(x)), _ptr(_h(out))
    for j, v in enumerate([planes, Hin, Win, Hout, Wout]):
        op.i[j] = int(v)
    return op, (x, out)


def minmax(x, scratch, out, *, n, nb=256):
    """{min, max} of the first n halfs of x -> out float[2] on the device; scratch float[2*nb]."""
    assert scratch.dtype == torch.float32 and scratch.numel() >= 2 * nb and out.dtype == torch.float32 and out.numel() >= 2
    op = L2dOp()
    op.kind = _lib.OP_MINMAX
    op.p[0], op.p[1], op.p[2] = _ptr(_h(x)), _ptr(scratch), _ptr(out)
    op.l[0] = int(n)
    op.i[0] = int(nb)
    return op, (x, scratch, out)


def depth_norm_resize(depth, mm, out, *, B, Hd, Wd, H, W):
    """reference pipeline_stream_animation_depth.py:560-567 in one pass: depth [B,Hd,Wd] fp16 + {min,max} -> [B,3,H,W] fp16."""
    assert mm.dtype == torch.float32
    op = L2dOp()
    op.kind = _lib.OP_DEPTH_NORM_RESIZE
    op.p[0], op.p[1], op.p[2] = _ptr(_h(depth)), _ptr(mm), _ptr(_h(out))
    for j, v in enumerate([B, Hd, Wd, H, W]):
        op.i[j] = int(v)
    return op, (depth, mm, out)


def lcm_step(x, eps, scal, x0, *, N, per):
    op = L2dOp()
    op.kind = _lib.OP_LCM_STEP
    op.p[0], op.p[1], op.p[2], op.p[3] = _ptr(_h(x)), _ptr(_h(eps)), _ptr(scal), _ptr(_h(x0))
    op.i[0], op.i[1] = N, per
    return op, (x, eps, scal, x0)


def ring_update(bias, pe_idx, update_idx, *, N, L, sink, frame_ctr=None):
    """update_attn_bias on the device, in place (reference pipeline_stream_animation_depth.py:416-438)."""
    assert bias.dtype == torch.float16 and pe_idx.dtype == torch.int64 and update_idx.dtype == torch.int64
    assert frame_ctr is None or (frame_ctr.dtype == torch.int64 and frame_ctr.numel() >= 1)
    op = L2dOp()
    op.kind = _lib.OP_RING_UPDATE
    op.p[0], op.p[1], op.p[2], op.p[3] = _ptr(bias), _ptr(pe_idx), _ptr(update_idx), _ptr(frame_ctr)
    op.i[0], op.i[1], op.i[2] = N, L, sink
    return op, (bias, pe_idx, update_idx, frame_ctr)


def stream_shift(x_t, eps, scal, x0_out, *, N, per, noise=None, depth=None):
    """LCM step of the N rows + stream-batch shift register, in place on x_t / depth (reference :387-401, :590-601)."""
    assert scal.dtype == torch.float32 and scal.numel() >= 4 * N
    assert noise is None or noise.numel() >= (N - 1) * per
    op = L2dOp()
    op.kind = _lib.OP_STREAM_SHIFT
    op.p[0], op.p[1], op.p[2], op.p[3] = _ptr(_h(x_t)), _ptr(_h(eps)), _ptr(scal), _ptr(noise)
    op.p[4], op.p[5] = _ptr(_h(x0_out)), _ptr(depth)
    op.i[0], op.i[1] = N, per
    return op, (x_t, eps, scal, noise, x0_out, depth)


def randn(out, *, seed: int, offset: int = 0, frame_ctr=None):
    """Standard-normal fill (Philox4x32-10 + Box-Muller); with `frame_ctr` the stream position advances per frame."""
    op = L2dOp()
    op.kind = _lib.OP_RANDN
    op.p[0], op.p[1] = _ptr(_h(out)), _ptr(frame_ctr)
    op.l[0], op.l[1], op.l[2] = out.numel(), int(seed), int(offset)
    return op, (out, frame_ctr)


def copy(src, dst, nbytes):
    op = L2dOp()
    op.kind = _lib.OP_COPY
    op.p[0], op.p[1] = _ptr(src), _ptr(dst)
    op.l[0] = int(nbytes)
    return op, (src, dst)


def run(op_and_keep, stream=None):
    """Run a single op immediately (unit tests)."""
    op, keep = op_and_keep
    pl = _lib.OpList()
    pl.append(op, *keep)
    pl.run(stream)
