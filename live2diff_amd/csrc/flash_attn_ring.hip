// Flash attention on MFMA for gfx950, LDS-DMA ring version: spatial self-attention (T x T, T up to 9216) and text
// cross-attention (T x 77) of the SD-1.5 transformer blocks -- the SDPA call inside diffusers' `Attention` that the
// reference reaches from attention.py:243 (attn1) and :250-255 (attn2).
//
//   O[q][:] = softmax_k( Q[q].K[k] / sqrt(d) ) . V[k][:]        per (batch b, head h), no mask
//
// Same "swapped" formulation as flash_attn.hip (S^T = K.Q^T, O^T += V^T.P^T with v_mfma_f32_16x16x32_f16: the S^T
// accumulator layout IS a legal k-slot assignment of the second MFMA's B operand, so P never leaves registers).  What
// changed against that first kernel, which the round-1 PMC pass showed at 21.7 % MFMA-busy with 27 % of its LDS cycles
// lost to bank conflicts and every K / V^T byte staged through VGPRs one tile ahead:
//   * K and V^T tiles go HBM/L2 -> LDS with `global_load_lds_dwordx4` (16 B per lane, no VGPR round trip) into an NS-stage
//     ring; one raw `s_barrier` per key tile and a counted `s_waitcnt vmcnt((NS-2) * LPS)`: NS-1 tiles are always in
//     flight.  The old kernel prefetched ONE tile through registers: with one block per CU (T = 1024 / 256 levels) a tile
//     took as long as a memory round trip (~1.8 us for ~0.3 us of MFMA work).
//   * The LDS image of a DMA is lane-linear (wave-uniform base + 16 * lane), so rows cannot be padded; the bank swizzle is
//     applied on the SOURCE side instead (a lane fetches the logical 16-byte slot  p ^ swz(row)  of its row) and again on the
//     fragment reads.  K is kept as KK sub-tiles of [64 keys][32 halfs] (64-byte rows, slot ^ ((row>>1)&3): the igemm BK = 32
//     image, conflict-free for the ds_read_b128 lane groups); V^T as [d rows][64 keys] (128-byte rows, slot ^ ((row>>1)&7):
//     conflict-free for the 16-byte fragment reads).  Zero padding (d = 40 -> 64 in QK^T, -> 48 rows in PV) and the
//     row of ones that yields the softmax denominator on the matrix cores are written ONCE per ring slot; the DMA never
//     touches them (exec-masked lanes), so padding costs no memory traffic.
//   * Softmax: Q is pre-scaled by log2(e)/sqrt(d) when it is loaded (fp16, one more rounding of the size Q already
//     carries) and the running reference -m is the accumulator INPUT of the first QK^T MFMA, so the MFMA result is already
//     s*c - m: a probability is ONE v_exp_f32 (the old kernel: FMA + exp).  The reference is only raised -- and O, l rescaled,
//     in a wave-uniform branch, with the pending tile's scores adjusted before they are exponentiated -- when some row
//     exceeds it by more than 2^8 (p <= 256 is harmless in fp16; O, l are fp32).
//   * Key order inside a tile is free as long as K rows and V^T columns agree.  The S^T accumulator gives lane group g the
//     tile rows {16 ks + 4 g + r}; the K rows are therefore DMA'd in the permuted order  row 32c + 16h + 4g + r <- key
//     32c + 8g + 4h + r, which makes the 8 keys a lane needs for one PV MFMA CONTIGUOUS in V^T: one conflict-free
//     ds_read_b128 per fragment instead of two 8-byte reads (which hipcc fused into ds_read2st64_b64: half rate and 2-way
//     bank conflicts -- 30 % of the LDS cycles of the first version of this kernel, PMC pass profiles/r2d_pmc_ops.txt).
//   * Query rows per wave are a template parameter (32 or 16): levels with few queries (T <= 1024) get twice the blocks.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// analysis builds of the pipelined loop with one component removed (tools/flash_ablate.sh; results are NOT attention):
// -DFAR_X=mask: 1 no lane maxima / test, 2 exponentials -> multiplies, 4 no refill DMA in the steady loop, 8 no barrier there,
// 16 no fragment reads there, 32 no PV MFMAs, 64 no QK^T MFMAs
#ifndef FAR_X
#define FAR_X 0
#endif
// schedule choices of the pipelined loop (A/B'd with tools/flash_ablate.sh): 1 PV MFMAs alternate between the two 16-row subtiles
// (dependent MFMAs six apart instead of three) with every exponential beside QK^T, 2 the refill DMA is issued piece by piece
// beside the PV MFMAs instead of behind the barrier, 4 s_setprio 1 over the two MFMA phases
#ifndef FAR_V
#define FAR_V 0
#endif

#define L2D_GPTR(p) ((__attribute__((address_space(1))) const void *)(p))
#define L2D_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

struct FARArgs {
    const h16 *q, *k, *vt, *zero;
    h16 *out;
    int B, H, d, Tq, Tk, ldq, ldk, ldvt, ldo, xcd;
    long long sq, sk, svt, so;
#ifdef L2D_PROBES
    unsigned long long *probe;   // analysis builds: s_memtime stamps of key tiles 8 and 9, every wave (tools/flash_probe.py)
#endif
};

#ifdef L2D_PROBES
static unsigned long long *g_flash_probe = nullptr;
extern "C" void l2d_flash_set_probe(void *p) { g_flash_probe = (unsigned long long *)p; }
// stamp i (0..7) of this wave: [block][wave][8]
#define FAR_STAMP(kt, i)                                                                                         \
    do {                                                                                                         \
        if (a.probe && (kt) == 8 && (threadIdx.x & 63) == 0)                                                      \
            a.probe[((unsigned long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define FAR_STAMP(kt, i) do { } while (0)
#endif

// max over the 4 lanes {li, li+16, li+32, li+48} that share a query row, without the LDS round trips of ds_bpermute:
// v_permlane16_swap / v_permlane32_swap exchange half-rows / half-waves between two registers
__device__ __forceinline__ float far_row_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int D, int NW = 4, int NSF = 0>     // NW waves per block (4 or 8); NSF forces the ring depth (0 = by LDS size)
struct FARCfg {
    static constexpr int KK = (D + 31) / 32;              // [64][32] K sub-tiles (QK^T contraction steps)
    static constexpr int D16 = (D + 15) / 16;             // 16-row blocks of O^T
    static constexpr int DV = D16 * 16;                   // V^T rows in LDS
    static constexpr bool ONES = (D % 16) != 0;           // a spare padded row of V^T carries ones -> softmax denominator
    static constexpr int KBYTES = KK * 64 * 32 * 2;
    static constexpr int VBYTES = DV * 64 * 2;
    static constexpr int STAGE = KBYTES + VBYTES;         // bytes
    static constexpr int NS = NSF ? NSF : ((STAGE * 4 <= 60 * 1024) ? 4 : 3);
    static constexpr int NKI = KK * 4;                    // K DMA wave-instructions per tile (16 rows of one sub-tile each)
    static constexpr int KPW = (NKI + NW - 1) / NW;       // ... per wave (waves without a real one issue a dummy)
    static constexpr int NVI = (D + 7) / 8;               // V^T DMA wave-instructions per tile (8 rows each)
    static constexpr int VPW = (NVI + NW - 1) / NW;       // ... per wave
    static constexpr int LPS = KPW + VPW;                 // DMA instructions per wave per stage
    static constexpr int LDS = NS * STAGE + 1024;         // + 1 KB landing zone for the dummy DMAs
};

template <int D, int QS, int NW, int NSF, int PIPE>
__device__ __forceinline__ void flash_ring_body(const FARArgs &a) {
    using Cf = FARCfg<D, NW, NSF>;
    constexpr int KK = Cf::KK, D16 = Cf::D16, DV = Cf::DV, NS = Cf::NS, LPS = Cf::LPS, VPW = Cf::VPW, NVI = Cf::NVI;
    constexpr int KPW = Cf::KPW, NKI = Cf::NKI, NT = 64 * NW;
    constexpr int STAGE_H = Cf::STAGE / 2, KH = Cf::KBYTES / 2;      // halfs
    extern __shared__ __attribute__((aligned(16))) h16 smem[];      // the ONLY LDS object: ring, then the dummy zone

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // XCD-aware block order.  Blocks are dispatched round-robin over the 8 XCDs (private L2s): with (q-block, head, batch)
    // as a plain 3-D grid every XCD would see the K / V of EVERY head (10.5 MB at cfg-2, 32 q-blocks re-reading each head's
    // 655 KB) and thrash its 4 MB L2.  The 1-D grid is remapped (bijectively) so that each XCD runs a contiguous range of
    // (batch, head, q-block): the q-blocks of a head share one L2, which then holds just ~2 heads.
    const int nqb = (a.Tq + 16 * NW * QS - 1) / (16 * NW * QS);
    int wgid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wgid = a.xcd ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : (int)blockIdx.x;
    }
    const int bh = wgid / nqb, qb = wgid - bh * nqb;
    const int b = bh / a.H, h = bh - b * a.H;
    const int q0 = qb * (16 * NW * QS) + wave * (16 * QS);
    const h16 *qp = a.q + (long long)b * a.sq + h * D;
    const h16 *kp = a.k + (long long)b * a.sk + h * D;
    const h16 *vp = a.vt + (long long)b * a.svt + (long long)h * D * a.ldvt;
    h16 *op = a.out + (long long)b * a.so + h * D;
    const int nt = (a.Tk + 63) / 64;

    // ---- Q rows: requested FIRST, all of them at once and unconditionally (clamped row / column: the selects come after the data),
    // so that their round trip runs under the LDS initialisation below.  (Until round 6 each of the QS x KK loads sat behind its own
    // bounds branch and hipcc waited `vmcnt(0)` after every one: four serialised cold round trips in front of the first DMA --
    // 2-4 us of a 10-13 us launch at the few-token levels and on the text keys.)
    h16x8 qraw[QS][KK];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int qr = min(q0 + qs * 16 + li, a.Tq - 1), dc = kk * 32 + lg * 8;
            qraw[qs][kk] = l2d_ld8(qp + (long long)qr * a.ldq + (dc < D ? dc : 0));
        }
    __builtin_amdgcn_sched_barrier(0);

    // ---- static LDS content, written once: zeros everywhere (QK^T / PV padding), ones in V^T row D of every stage
    {
        const h16x8 z = l2d_zero8();
        for (int i = tid; i < (Cf::LDS / 16); i += NT) l2d_st8(smem + i * 8, z);
        if (Cf::ONES) {
            __syncthreads();
            h16x8 one;
#pragma unroll
            for (int e = 0; e < 8; ++e) one[e] = (h16)1.0f;
            if (tid < NS * 8) l2d_st8(smem + (tid >> 3) * STAGE_H + KH + D * 64 + (tid & 7) * 8, one);   // a full row: swizzle-invariant
        }
    }

    // ---- Q fragments (B operand), pre-scaled by log2(e) / sqrt(d): lane (j = li, g = lg) holds Q[q0 + qs*16 + j][kk*32 + 8g ..]
    const float c2e = rsqrtf((float)D) * 1.4426950408889634f;
    h16x8 qf[QS][KK];
    {
        h16x8 sc;
#pragma unroll
        for (int e = 0; e < 8; ++e) sc[e] = (h16)c2e;
#pragma unroll
        for (int qs = 0; qs < QS; ++qs)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int qr = q0 + qs * 16 + li, dc = kk * 32 + lg * 8;
                qf[qs][kk] = (qr < a.Tq && dc < D) ? qraw[qs][kk] * sc : l2d_zero8();
            }
    }

    // ---- DMA descriptors (tile-invariant parts).  K sub-tile kk: this wave fills rows 16w .. 16w+15, lane l = (row l>>2, physical slot l&3).
    // K piece p = wave + NW * i (i < KPW): sub-tile p / 4, rows 16 * (p % 4) ..; with 4 waves a wave fills the same 16 rows of
    // every sub-tile, with 8 waves (d <= 64) each wave fills one piece.
    const int krow = ((wave + 0) & 3) * 16 + (lane >> 2);                 // NW is 4 or 8: p % 4 == wave % 4 for every i
    const int kcol = (((lane & 3) ^ ((krow >> 1) & 3)) << 3);             // logical column (halfs) inside the sub-tile
    // tile row krow = 32c + 16h + 4g + r holds key 32c + 8g + 4h + r (see the header): PV fragments become 16-byte reads
    const int kkey = (krow & 32) | (((krow >> 2) & 3) << 3) | (((krow >> 4) & 1) << 2) | (krow & 3);
    // V^T: wave-instruction j fills rows 8j .. 8j+7, lane l = (row l>>3, physical slot l&7); this wave owns j = wave, wave+4, ...
    int vrow[VPW], vkey[VPW];
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
        const int j = wave + NW * i;
        vrow[i] = j * 8 + (lane >> 3);
        vkey[i] = (((lane & 7) ^ ((vrow[i] >> 1) & 7)) << 3);             // first key (within the tile) of the logical slot
    }
    h16 *dummy = smem + NS * STAGE_H;
    // Per-lane source pointers of tile 0 and their per-tile strides.  The loop is bound by instruction issue, so the DMA side
    // is kept to "pointer += stride; load" per piece: lanes that only ever fetch padding (d = 40: columns 40..63 of the second
    // K sub-tile) point at the zero page with stride 0 instead of being exec-masked (no divergent branches around the DMAs),
    // and the key < Tk test exists only in the code path of the last tile.
    const h16 *kptr[KPW];
    long long kadv[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int pc = wave + NW * i, kk = pc >> 2;
        const bool real = pc < NKI && kk * 32 + kcol < D;
        kptr[i] = real ? kp + (long long)kkey * a.ldk + kk * 32 + kcol : a.zero;
        kadv[i] = real ? 64ll * a.ldk : 0ll;
    }
    const h16 *vptr[VPW];
    long long vadv[VPW];
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
        const bool real = (wave + NW * i) < NVI && vrow[i] < D;
        vptr[i] = real ? vp + (long long)vrow[i] * a.ldvt + vkey[i] : a.zero;
        vadv[i] = real ? 64ll : 0ll;
    }

    int is_slot = 0, is_key0 = 0;
    auto issue_piece = [&](int p, auto last_tag) {                        // DMA wave-instruction p (0 .. LPS - 1) of a stage
        constexpr bool LAST = decltype(last_tag)::value;                  // only the last tile can reach beyond Tk
        h16 *st = smem + is_slot * STAGE_H;
        if (p < KPW) {
            const int i = p, pc = wave + NW * i;                          // wave-uniform piece index
            const h16 *src = (LAST && is_key0 + kkey >= a.Tk) ? a.zero : kptr[i];
            h16 *dst = pc < NKI ? st + (pc >> 2) * 2048 + (pc & 3) * 512 : dummy;
            __builtin_amdgcn_global_load_lds(L2D_GPTR(src), L2D_LPTR(dst), 16, 0, 0);
            kptr[i] += kadv[i];
        } else {
            const int i = p - KPW, j = wave + NW * i;                     // wave-uniform
            const h16 *src = (LAST && is_key0 + vkey[i] >= a.Tk) ? a.zero : vptr[i];
            h16 *dst = j < NVI ? st + KH + j * 512 : dummy;
            __builtin_amdgcn_global_load_lds(L2D_GPTR(src), L2D_LPTR(dst), 16, 0, 0);
            vptr[i] += vadv[i];
        }
    };
    auto issue_end = [&]() {
        is_slot = (is_slot + 1 == NS) ? 0 : is_slot + 1;
        is_key0 += 64;
    };
    auto issue = [&](auto last_tag) {                                     // LPS DMA wave-instructions, always
#pragma unroll
        for (int p = 0; p < LPS; ++p) issue_piece(p, last_tag);
        issue_end();
    };

    // fragment read offsets (halfs, relative to the stage base); the swizzle terms are tile- and block-row-invariant
    const int kswz = (li >> 1) & 3, vswz = (li >> 1) & 7;
    const int koff = li * 32 + ((lg ^ kswz) << 3);                        // + kk*2048 + ks*512
    int voff[2];                                                          // [c2]: + ds*1024; keys 32 c2 + 8 lg .. + 7
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) voff[c2] = KH + li * 64 + (((c2 * 4 + lg) ^ vswz) << 3);

    f32x4 oacc[D16][QS];
#pragma unroll
    for (int ds = 0; ds < D16; ++ds)
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) oacc[ds][qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mref[QS], lrow[QS];
    f32x4 cinit[QS];                                                      // -mref, the accumulator input of the first QK^T MFMA
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) { mref[qs] = 0.f; lrow[qs] = 0.f; cinit[qs] = (f32x4){0.f, 0.f, 0.f, 0.f}; }


    using T_ = std::true_type;
    using F_ = std::false_type;

    // V^T columns in [Tk, round_up(Tk, 8)) of the last tile may hold anything (the producer pads rows to 16 bytes): P is
    // exactly 0 there, but 0 * NaN is not.  They are cleared in LDS after the tile has landed.
    auto scrub_last = [&]() {
        const int first = a.Tk & 63, last = ((a.Tk + 7) & ~7) & 63;      // key columns inside the last tile
        if ((a.Tk & 7) == 0) return;
        h16 *vs = smem + ((nt - 1) % NS) * STAGE_H + KH;
        for (int r = tid; r < D; r += NT) {
            const int sw = (r >> 1) & 7;
            for (int c = first; c < (last == 0 ? 64 : last); ++c) vs[r * 64 + ((((c >> 3) ^ sw)) << 3) + (c & 7)] = (h16)0.0f;
        }
        __syncthreads();
    };

    if constexpr (PIPE != 0) {
        // ---- software-pipelined loop (round 6; variant 4).  One wave's instruction stream of the loop above is four serial
        // segments -- K fragments + QK^T (matrix pipe only), row maxima with two cross-lane exchanges per 16-row subtile (VALU
        // only, a dependent chain), exponentials (VALU only), PV (matrix pipe only) -- and its partner wave on the SIMD runs the
        // same segments: stamps (profiles/r3m_flash_tile_phases.txt) show 2 344 cycles per tile against 448 of either pipe.
        // Here every segment pairs matrix work with INDEPENDENT vector work of the neighbouring tile, in one wave:
        //   phase A:  S(t+1) = K(t+1).Q^T  (16 MFMAs)   beside   P(t) = exp2(S(t)) -> fp16   (32 v_exp + 16 v_cvt_pk)
        //   phase B:  O += V^T(t).P(t)     (12 MFMAs)   beside   the reference test of S(t+1)
        // and the test no longer reduces across lanes: a row exceeds the reference by 2^8 iff SOME lane's partial maximum
        // does, so the fast path is 16 v_max3 + one compare (no permlane, no dependent exchange); the full row maxima are
        // taken only inside the rarely taken rescale branch.  K fragments of tile t+1 and V^T fragments of tile t are read at
        // the top of the iteration (tile t+1 has landed one iteration early: the ring holds t, t+1 and two tiles in flight),
        // so no LDS latency sits between a barrier and the first MFMA.  Arithmetic and its order are those of the loop
        // above: the outputs are bit-identical to variant 2 (asserted in tests/test_gpu_kernels.py).
        static_assert(QS == 2 && KK <= 2 && D16 <= 3 && NS == 4, "pipelined body: 32 query rows per wave, d <= 48, 4-slot ring");
        constexpr int EPS = 4 / KK;                                       // exponentials per QK^T MFMA (32 / (8 KK))
        constexpr int NPV = 4 * D16;                                      // PV MFMAs per tile
        f32x4 sA[4][2], sB[4][2];
        h16x8 kf[KK][4], vf[2][D16], pf[2][2];

        auto ld_k = [&](int slot) {
            const h16 *st = smem + slot * STAGE_H;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[kk][ks] = l2d_ld8(st + kk * 2048 + ks * 512 + koff);
        };
        auto ld_v = [&](int slot) {
            const h16 *st = smem + slot * STAGE_H;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int ds = 0; ds < D16; ++ds) vf[c2][ds] = l2d_ld8(st + ds * 1024 + voff[c2]);
        };
        auto qk_mfma = [&](f32x4 (&S)[4][2], int e) {                     // MFMA e (0 .. 8 KK - 1) of a tile's QK^T, subtile-major
            const int qs = e / (4 * KK), kk = (e / 4) % KK, ks = e % 4;
            if (!(FAR_X & 64)) S[ks][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kk][ks], qf[qs][kk], kk == 0 ? cinit[qs] : S[ks][qs], 0, 0, 0);
            else S[ks][qs][0] = (float)kf[kk][ks][0] + (kk == 0 ? cinit[qs][0] : S[ks][qs][0]);
        };
        auto mask_tail = [&](f32x4 (&S)[4][2], int kt) {                  // keys beyond Tk exist in the last tile only
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((kt * 64 + (ks >> 1) * 32 + lg * 8 + (ks & 1) * 4 + r) >= a.Tk) S[ks][qs][r] = -3.0e38f;
        };
        auto row_max = [&](f32x4 (&S)[4][2], float (&mx)[2]) {
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                float m = fmaxf(fmaxf(S[0][qs][0], S[0][qs][1]), fmaxf(S[0][qs][2], S[0][qs][3]));
#pragma unroll
                for (int ks = 1; ks < 4; ++ks)
                    m = fmaxf(fmaxf(m, fmaxf(S[ks][qs][0], S[ks][qs][1])), fmaxf(S[ks][qs][2], S[ks][qs][3]));
                mx[qs] = far_row_max(m);
            }
        };
        // Raise the reference: everything still at the old one -- O, l and the not yet exponentiated scores S -- is moved to
        // the new one exactly once (the PV MFMAs of the previous tile are program-order BEFORE this, so O is complete).
        auto rescale = [&](f32x4 (&S)[4][2], auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            float mx[2];
            row_max(S, mx);
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
                const float delta = FIRST ? mx[qs] : fmaxf(mx[qs], 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                mref[qs] += delta;
                lrow[qs] *= alpha;
#pragma unroll
                for (int ds = 0; ds < D16; ++ds) oacc[ds][qs] *= alpha;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) S[ks][qs] -= delta;
                cinit[qs] = (f32x4){-mref[qs], -mref[qs], -mref[qs], -mref[qs]};
            }
        };
        float psum[2] = {0.f, 0.f};
        auto exp_slice = [&](f32x4 (&S)[4][2], int x) {                   // exponential x (0 .. 31) of a tile, subtile-major
            const int qs = x / 16, ks = (x % 16) / 4, r = x % 4;
            const float pv = (FAR_X & 2) ? S[ks][qs][r] * 0.5f : __builtin_amdgcn_exp2f(S[ks][qs][r]);
            if (!Cf::ONES) psum[qs] += pv;
            pf[ks >> 1][qs][(ks & 1) * 4 + r] = (h16)pv;
        };
        auto pv_mfma = [&](int e) {                                       // MFMA e (0 .. NPV - 1) of a tile's PV, subtile-major
            const int qs = (FAR_V & 1) ? (e / D16) % 2 : e / (2 * D16), c2 = (FAR_V & 1) ? e / (2 * D16) : (e / D16) % 2, ds = e % D16;
            if (!(FAR_X & 32)) oacc[ds][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[c2][ds], pf[c2][qs], oacc[ds][qs], 0, 0, 0);
            else oacc[ds][qs][0] += (float)pf[c2][qs][0] + (float)vf[c2][ds][0];
        };
        // lane-local maximum of a tile's 32 scores: 16 three-input maxima in two chains (one per subtile), op o = 0 .. 15
        float lm[2];
        auto lane_max_op = [&](f32x4 (&S)[4][2], int o) {
            const int qs = o / 8, st = o % 8;                            // step 0: values 0..2; step k: values 2k+1, 2k+2; step 7: value 15
            auto val = [&](int i) -> float { return S[i / 4][qs][i % 4]; };
            if (st == 0) lm[qs] = fmaxf(fmaxf(val(0), val(1)), val(2));
            else if (st < 7) lm[qs] = fmaxf(fmaxf(lm[qs], val(2 * st + 1)), val(2 * st + 2));
            else lm[qs] = fmaxf(lm[qs], val(15));
        };

        // The schedule is pinned: between two `sched_barrier(0)` hipcc may order instructions, across one it may not (left to
        // itself it hoists the maxima to just behind the MFMAs that produce their operands -- a dependency stall each -- and
        // issues the twelve PV MFMAs back to back with no vector work beside them).  One slot = one MFMA + the vector work
        // that runs in its shadow (the matrix pipe is busy for 16 cycles per MFMA, a plain VALU issue takes 2, v_exp ~5-8).
        //   exponentials as 16 pairs (one v_cvt_pk each): pairs 0-1 ahead of the first MFMA, over the latency of the K
        //   fragment reads; pairs 2-13 beside QK^T MFMAs; pairs 14-15 (last of subtile 1) beside the first PV MFMAs of
        //   subtile 0; the lane maxima of S(t+1) beside the rest of PV, well behind the QK^T MFMAs they read.
#define FAR_SB() __builtin_amdgcn_sched_barrier(0)
        constexpr int NQK = 8 * KK;
        constexpr int NPA = (FAR_V & 1) ? 14 : 12;                        // exponential pairs beside the QK^T MFMAs (after the two up front)
        auto exp_pair = [&](f32x4 (&S)[4][2], int pr) { exp_slice(S, 2 * pr); exp_slice(S, 2 * pr + 1); };
        auto phase_a = [&](f32x4 (&Sc)[4][2], f32x4 (&Sn)[4][2], int vslot) {
            exp_pair(Sc, 0);
            exp_pair(Sc, 1);
            FAR_SB();
            if (!(FAR_X & 16)) ld_v(vslot);
            FAR_SB();
            // pairs 2 .. 13 over the NQK slots
#pragma unroll
            for (int e = 0; e < NQK; ++e) {
                qk_mfma(Sn, e);
#pragma unroll
                for (int pr = 2 + (e * NPA) / NQK; pr < 2 + ((e + 1) * NPA) / NQK; ++pr) exp_pair(Sc, pr);
                FAR_SB();
            }
        };
        auto phase_b = [&](f32x4 (&Sc)[4][2], f32x4 (&Sn)[4][2], auto dma_tag) {
            constexpr bool DMA = decltype(dma_tag)::value;                // steady iterations with FAR_V & 2: the refill rides here
#pragma unroll
            for (int e = 0; e < NPV; ++e) {
                pv_mfma(e);
                if (!(FAR_V & 1) && e == 0) exp_pair(Sc, 14);
                if (!(FAR_V & 1) && e == 1) exp_pair(Sc, 15);
                constexpr int E0 = (FAR_V & 1) ? 0 : 2;
                if (e >= E0 && !(FAR_X & 1)) {
#pragma unroll
                    for (int o = ((e - E0) * 16) / (NPV - E0); o < ((e - E0 + 1) * 16) / (NPV - E0); ++o) lane_max_op(Sn, o);
                }
                if (DMA) {
#pragma unroll
                    for (int p = 0; p < LPS; ++p)
                        if (e == ((2 * p + 1) * NPV) / (2 * LPS)) issue_piece(p, F_{});
                }
                FAR_SB();
            }
            if (DMA) issue_end();
            if (!Cf::ONES) { lrow[0] += psum[0]; lrow[1] += psum[1]; psum[0] = 0.f; psum[1] = 0.f; }
            if (!(FAR_X & 1) && __builtin_expect(__any(fmaxf(lm[0], lm[1]) > 8.0f), 0)) {
                asm volatile("" ::: "memory");   // volatile: the rarely needed rescale arithmetic must not be speculated into the hot path
                rescale(Sn, F_{});
            }
        };

        // one iteration: tile t is exponentiated and multiplied into O, tile t + 1 gets its scores
        auto iter = [&](int t, f32x4 (&Sc)[4][2], f32x4 (&Sn)[4][2], auto steady_tag) {
            constexpr bool STEADY = decltype(steady_tag)::value;          // t + 3 < nt - 1: refill without bounds tests
            FAR_STAMP(t, 4);
            FAR_STAMP(t - 1, 7);
            if (STEADY) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory"); // own share of tile t + 1 landed (t + 2 in flight)
                FAR_STAMP(t, 5);
                if (!(FAR_X & 8)) __builtin_amdgcn_s_barrier();           // tile t + 1 complete; everyone is done with tile t - 1
                FAR_STAMP(t, 6);
                if (!(FAR_X & 4) && !(FAR_V & 2)) issue(F_{});            // tile t + 3 into the slot of tile t - 1
            } else {
                if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (t + 1 == nt - 1) scrub_last();
                if (t + 3 < nt - 1) issue(F_{});
                else if (t + 3 == nt - 1) issue(T_{});
            }
            FAR_STAMP(t, 0);
            if (!(STEADY && (FAR_X & 16))) ld_k((t + 1) & 3);
            FAR_SB();
            FAR_STAMP(t, 2);
            if (FAR_V & 4) __builtin_amdgcn_s_setprio(1);
            phase_a(Sc, Sn, t & 3);
            FAR_STAMP(t, 1);
            if (!STEADY && t + 1 == nt - 1 && nt * 64 > a.Tk) mask_tail(Sn, t + 1);
            if (STEADY && (FAR_V & 2) && !(FAR_X & 4)) phase_b(Sc, Sn, T_{});
            else phase_b(Sc, Sn, F_{});
            if (FAR_V & 4) __builtin_amdgcn_s_setprio(0);
            FAR_STAMP(t, 3);
        };
        auto tail = [&](f32x4 (&Sc)[4][2]) {                              // last tile: exponentials and PV only
            ld_v((nt - 1) & 3);
#pragma unroll
            for (int x = 0; x < 32; ++x) exp_slice(Sc, x);
#pragma unroll
            for (int e = 0; e < NPV; ++e) pv_mfma(e);
            if (!Cf::ONES) { lrow[0] += psum[0]; lrow[1] += psum[1]; }
        };

        __syncthreads();                                                  // static LDS content in place before any DMA lands
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nt) {
                if (s == nt - 1) issue(T_{});
                else issue(F_{});
            }
        if (nt >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        else if (nt == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nt == 1) scrub_last();
        ld_k(0);
#pragma unroll
        for (int e = 0; e < 8 * KK; ++e) qk_mfma(sA, e);
        if (nt == 1 && 64 > a.Tk) mask_tail(sA, 0);
        rescale(sA, T_{});                                                // tile 0 fixes the reference

        int t = 0;
        for (; t + 6 <= nt - 1; t += 2) {                                 // t + 1 <= nt - 5: both refills are plain
            iter(t, sA, sB, T_{});
            iter(t + 1, sB, sA, T_{});
        }
        for (; t + 2 <= nt - 1; t += 2) {
            iter(t, sA, sB, F_{});
            iter(t + 1, sB, sA, F_{});
        }
        if (t == nt - 2) {
            iter(t, sA, sB, F_{});
            tail(sB);
        } else {
            tail(sA);
        }
    } else {
    int cp_slot = 0;
    // FIRST: tile 0 fixes the reference (no previous one); LAST: the only tile that can hold keys >= Tk.  Both are
    // compile-time so the steady-state body carries neither the masking selects nor the first-tile test.
    auto compute = [&](int kt, auto first_tag, auto last_tag) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        const h16 *st = smem + cp_slot * STAGE_H;
        // Instruction order is chosen so that ONE wave keeps both pipes busy (the matrix pipe executes an MFMA for 16
        // cycles after a 4-cycle issue; independent VALU work issued behind it runs in its shadow): q-subtile-major QK^T
        // (row maximum of subtile 0 under the MFMAs of subtile 1), one rescale decision for both subtiles (a single rarely
        // taken branch instead of two basic-block boundaries), then exp / pack of subtile 1 under the PV MFMAs of subtile 0.
        // (The subtile-major order holds all K / V^T fragments of a tile in registers: 32 + 24 VGPRs at d = 40.  The wide
        // heads would spill -- they keep the fragment-major order and rely on the second resident block for overlap.)
        constexpr bool SUBTILE_MAJOR = (QS == 2 && KK <= 2 && D16 <= 3);
        f32x4 sacc[4][QS];
        float mx[QS];
        if (SUBTILE_MAJOR) {
            h16x8 kf[KK][4];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kf[kk][ks] = l2d_ld8(st + kk * 2048 + ks * 512 + koff);
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        sacc[ks][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kk][ks], qf[qs][kk], kk == 0 ? cinit[qs] : sacc[ks][qs], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const h16x8 k1 = l2d_ld8(st + kk * 2048 + ks * 512 + koff);
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs)
                        sacc[ks][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qf[qs][kk], kk == 0 ? cinit[qs] : sacc[ks][qs], 0, 0, 0);
                }
        }
        FAR_STAMP(kt, 1);                                                 // QK^T issued
        if (LAST && (kt + 1) * 64 > a.Tk) {                               // keys beyond Tk exist in the last tile only
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int qs = 0; qs < QS; ++qs)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((kt * 64 + (ks >> 1) * 32 + lg * 8 + (ks & 1) * 4 + r) >= a.Tk) sacc[ks][qs][r] = -3.0e38f;
        }
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            float m = fmaxf(fmaxf(sacc[0][qs][0], sacc[0][qs][1]), fmaxf(sacc[0][qs][2], sacc[0][qs][3]));
#pragma unroll
            for (int ks = 1; ks < 4; ++ks)
                m = fmaxf(fmaxf(m, fmaxf(sacc[ks][qs][0], sacc[ks][qs][1])), fmaxf(sacc[ks][qs][2], sacc[ks][qs][3]));
            mx[qs] = far_row_max(m);
        }
        FAR_STAMP(kt, 2);                                                 // row maxima known
        // sacc already is  s*c - mref.  Raise the reference (rarely after the first tile): everything still at the old
        // reference -- O, l and THIS tile's not yet exponentiated scores -- is moved to the new one exactly once.  A subtile
        // whose maximum did not grow gets delta = 0, alpha = 1: the same code path, exact.
        if (FIRST || __builtin_expect(__any((QS == 2 ? fmaxf(mx[0], mx[QS - 1]) : mx[0]) > 8.0f), 0)) {
            asm volatile("" ::: "memory");   // volatile: the rarely needed rescale arithmetic must not be speculated into the hot path
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                const float delta = FIRST ? mx[qs] : fmaxf(mx[qs], 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);      // (first tile: O = l = 0, any finite alpha will do)
                mref[qs] += delta;
                lrow[qs] *= alpha;
#pragma unroll
                for (int ds = 0; ds < D16; ++ds) oacc[ds][qs] *= alpha;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) sacc[ks][qs] -= delta;
                cinit[qs] = (f32x4){-mref[qs], -mref[qs], -mref[qs], -mref[qs]};
            }
        }
        constexpr bool VF_UP_FRONT = SUBTILE_MAJOR;                       // 8 * D16 VGPRs of V^T fragments held across both subtiles
        h16x8 vf[2][VF_UP_FRONT ? D16 : 1];
        if (VF_UP_FRONT) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int ds = 0; ds < D16; ++ds) vf[c2][VF_UP_FRONT ? ds : 0] = l2d_ld8(st + ds * 1024 + voff[c2]);
        }
        h16x8 pf[2][QS];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            float psum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sacc[ks][qs][r]);
                    if (!Cf::ONES) psum += pv;
                    pf[ks >> 1][qs][(ks & 1) * 4 + r] = (h16)pv;
                }
            if (!Cf::ONES) lrow[qs] += psum;
            if (VF_UP_FRONT) {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int ds = 0; ds < D16; ++ds)
                        oacc[ds][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[c2][VF_UP_FRONT ? ds : 0], pf[c2][qs], oacc[ds][qs], 0, 0, 0);
            }
        }
        if (!VF_UP_FRONT) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int ds = 0; ds < D16; ++ds) {
                    const h16x8 v1 = l2d_ld8(st + ds * 1024 + voff[c2]);
#pragma unroll
                    for (int qs = 0; qs < QS; ++qs)
                        oacc[ds][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, pf[c2][qs], oacc[ds][qs], 0, 0, 0);
                }
        }
        FAR_STAMP(kt, 3);                                                 // exp + PV issued
        cp_slot = (cp_slot + 1 == NS) ? 0 : cp_slot + 1;
    };

    __syncthreads();                                                      // static LDS content in place before any DMA lands
    // prologue: NS-1 tiles in flight
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) {
            if (s == nt - 1) issue(T_{});
            else issue(F_{});
        }
    int kt = 0;
    if (NS - 1 < nt) {                     // tile 0 with a refill behind it
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        __builtin_amdgcn_s_barrier();
        if (NS - 1 == nt - 1) issue(T_{});
        else issue(F_{});
        compute(0, T_{}, F_{});
        kt = 1;
    }
    for (; kt + NS < nt; ++kt) {           // refills of tiles < nt - 1: no bounds test anywhere in this loop
        FAR_STAMP(kt, 4);                  // (loop top of tile 8; the same stamp taken for tile 9 closes the period)
        FAR_STAMP(kt - 1, 7);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        FAR_STAMP(kt, 5);                  // own DMA share landed
        __builtin_amdgcn_s_barrier();      // every wave's share of tile kt is in LDS; everyone finished tile kt-1
        FAR_STAMP(kt, 6);                  // barrier passed
        issue(F_{});                       // refill the ring slot tile kt-1 occupied
        FAR_STAMP(kt, 0);                  // DMA issued
        compute(kt, F_{}, F_{});
    }
    for (; kt + (NS - 1) < nt; ++kt) {     // the one refill that fetches the last tile
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(T_{});
        compute(kt, F_{}, F_{});
    }
    // drain: no more refills.  (Waiting for everything costs nothing here: at most NS-1 short tiles remain.)
    for (; kt < nt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt == nt - 1) {
            scrub_last();
            if (kt == 0) compute(kt, T_{}, T_{});
            else compute(kt, F_{}, T_{});
        } else if (kt == 0) {
            compute(kt, T_{}, F_{});
        } else {
            compute(kt, F_{}, F_{});
        }
    }

    }   // !PIPE

#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        float l;
        if (Cf::ONES) {
            // O^T row D (fragment D/16, lane group (D%16)/4, register D%4) holds the denominator of query li
            l = __shfl(oacc[D / 16][qs][D % 4], ((D % 16) / 4) * 16 + li, 64);
        } else {
            l = lrow[qs];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int qr = q0 + qs * 16 + li;
        if (qr >= a.Tq) continue;
#pragma unroll
        for (int ds = 0; ds < D16; ++ds) {
            const int dc = ds * 16 + lg * 4;
            if (dc >= D) continue;
            h16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (h16)(oacc[ds][qs][r] * inv);
            *reinterpret_cast<h16x4 *>(op + (long long)qr * a.ldo + dc) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Removed in round 3 (negative results, recorded in DESIGN.md section 3.2 with their profiles): a software-pipelined body that
// issued the QK^T MFMAs of tile t+1 inside the softmax of tile t (parity-green, 95.6 vs 80.9 us at T = 4096, d = 40,
// profiles/r3l_flash_pipe_time.txt: the bound is the SIMD's shared VALU -- 32 quarter-rate v_exp_f32 + ~120 VALU instructions
// against 448 MFMA cycles per wave and 64-key tile -- so stretching one wave's MFMAs over its own VALU work only lengthened its
// critical path), 8 waves x 16 rows and a 2-deep ring with 4-5 blocks per CU (81.0 / 90.1 vs 81.0 us).

// NW waves x (16 * QS) queries per block.  The SIMD issues at most one VALU and one MFMA per 4 cycles, from DIFFERENT
// waves: with two waves per SIMD the in-order streams (28 MFMA + ~140 VALU + waits per wave and tile at d = 40) leave both
// pipes idle half of the time (PMC: 43 % issuing, 29 % waiting on counters / barriers, 28 % on dependencies).  More, thinner
// waves per SIMD fill those slots: QS = 1 halves the registers (<= 128: four waves per SIMD), and either 8 waves share a
// block's K / V tiles (same L2 -> LDS traffic per query as 4 x 32) or the ring is made shallow so that 4 blocks fit a CU.
// The occupancy hint also keeps the accumulators out of the AGPR file.
template <int D, int QS, int NW, int NSF, int PIPE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(QS == 1 && D <= 80 ? 3 : 2))) void flash_ring_kernel(FARArgs a) {
    flash_ring_body<D, QS, NW, NSF, PIPE>(a);
}

template <int D, int QS, int NW, int NSF, int PIPE = 0>
static int launch_far_q(const FARArgs &a, hipStream_t s) {
    using Cf = FARCfg<D, NW, NSF>;
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (Cf::LDS > 65536 && !attr_done) {
        if (hipFuncSetAttribute((const void *)flash_ring_kernel<D, QS, NW, NSF, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS) == hipSuccess)
            attr_done = true;
        else
            (void)hipGetLastError();
    }
    dim3 grid(((a.Tq + 16 * NW * QS - 1) / (16 * NW * QS)) * a.H * a.B);      // 1-D: decoded XCD-aware in the kernel
    hipLaunchKernelGGL((flash_ring_kernel<D, QS, NW, NSF, PIPE>), grid, dim3(64 * NW), Cf::LDS, s, a);
    return L2D_OK;
}

// geometry: 0 auto; 2 = 4 waves x 32 rows; 3 = 4 waves x 16 rows; 4 = 4 waves x 32 rows, software-pipelined loop (d <= 48)
template <int D>
static int launch_far(const FARArgs &a, int geo, hipStream_t s) {
    if (geo == 0) {   // auto: 32 query rows per wave when that still gives >= 1.5 blocks per CU, else 16; the pipelined loop
        const long long big = (long long)((a.Tq + 127) / 128) * a.H * a.B;     // wherever it is built (d <= 48: 80 -> 72-75 us at
        geo = (big >= 384) ? (D <= 48 ? 4 : 2) : 3;                            // T = 4096, 9.5 -> 8.5 us on the text keys)
    }
    if constexpr (D <= 48) {
        if (geo == 4) return launch_far_q<D, 2, 4, 0, 1>(a, s);
    }
    if (geo == 4) geo = 2;
    if constexpr (D <= 80) {          // d = 160 with 32 query rows per wave does not fit the register file
        if (geo == 2) return launch_far_q<D, 2, 4, 0>(a, s);
    }
    return launch_far_q<D, 1, 4, 0>(a, s);
}

// called from l2d_launch_flash_attn (flash_attn.hip) after argument validation
int l2d_launch_flash_ring(const l2d_op *op, int geo, hipStream_t s) {
    FARArgs a;
    a.q = (const h16 *)op->p[0]; a.k = (const h16 *)op->p[1]; a.vt = (const h16 *)op->p[2]; a.out = (h16 *)op->p[3];
    a.zero = (const h16 *)op->p[4];
#ifdef L2D_PROBES
    a.probe = g_flash_probe;
#endif
    a.xcd = 1;                          // XCD-contiguous block order (the plain order measured the same: round 2)
    a.B = op->i[0]; a.H = op->i[1]; a.d = op->i[2]; a.Tq = op->i[3]; a.Tk = op->i[4];
    a.ldq = op->i[5]; a.ldk = op->i[6]; a.ldvt = op->i[7]; a.ldo = op->i[8];
    a.sq = op->l[0]; a.sk = op->l[1]; a.svt = op->l[2]; a.so = op->l[3];
    switch (a.d) {
        case 8: return launch_far<8>(a, geo, s);
        case 16: return launch_far<16>(a, geo, s);
        case 32: return launch_far<32>(a, geo, s);
        case 40: return launch_far<40>(a, geo, s);
        case 64: return launch_far<64>(a, geo, s);
        case 80: return launch_far<80>(a, geo, s);
        case 160: return launch_far<160>(a, geo, s);
    }
    return L2D_EINVAL;
}
