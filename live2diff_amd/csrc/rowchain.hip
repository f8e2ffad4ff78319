// Token-resident tail of a transformer block on gfx950 (round 5): ONE launch for what the reference runs as
//     h2 = to_out(a) + h1                       attention output projection + residual   (attention.py:243-258, motion_module.py:401-435)
//     h3 = FF2( GEGLU( LayerNorm(h2) ) ) + h2    feed-forward, diffusers FeedForward(geglu) (attention.py:204,262-268; motion_module.py:360,429-433)
//     out = proj_out(h3) + x                     block output projection + block residual (attention.py:125-133, motion_module.py:290-297)
// and what the plan ran as four launches (row GEMM, row GEMM with GEGLU, implicit GEMM, row GEMM: 8 + 31 + 20 + 12 us at
// C = 320, M = 8192) with three activation round trips through HBM between them (the 4C-wide GEGLU tensor alone is 21 MB out and
// 21 MB back in).  Everything here is row-local per token, so a block owns BM = 32 tokens for the whole chain:
//   * three LDS tiles in the row GEMM's activation layout (16-byte slot q of token r at ((q BM) + (r ^ 2 (q & 7))) 16 B: a tile is
//     at once the MFMA B operand of the next GEMM and, read 8 bytes at a time, the residual of an epilogue): X (the current GEMM
//     input, 20 KB), H (the residual stream h1 -> h2, 20 KB), G (the 4C-wide GEGLU output, 80 KB); no activation leaves the CU
//     between the attention output and the block output;
//   * the weights stream L2 -> VGPR in rowgemm.hip's fragment order (ops.pack_rowgemm: LayerNorm gamma / beta folded into the GEGLU
//     projection) through a register ring of 8 k steps per 32-row tile; every wave owns NT 32-column tiles of the 320 output columns
//     of a pass, a 4C-wide layer is 8 passes.  NT = 1, TEN waves (round 6, second session): with NT = 2 the block was five waves
//     and the SIMD that hosted two of them carried twice the MFMAs and twice the GELU / epilogue work of the others -- component
//     ablation (tools/rowchain_time.py over tools/variant_libs.sh builds, profiles/round6_o_rowchain_ablation.txt: no weight loads
//     at all 45.6 -> 33.6 us, GELU -> identity -5.8, deeper ring +-0) showed the launch paced by that SIMD, not by the weight
//     ingest; ten one-tile waves sit 3 / 3 / 2 / 2 on the SIMDs: tail 45.7 -> 39.9 us, head segments 17.9 -> 16.1 us, bit-identical.  The whole chain is straight-line code (C is a template parameter: 280 k steps), so
//     hipcc counts every fragment exactly and the NEXT pass's first ring is requested in front of the CURRENT pass's epilogue --
//     the epilogues (bias, GELU, LDS traffic) run in the shadow of that round trip.  Biases live in LDS (staged once): a global
//     load in an epilogue would sit behind the ring in the in-order VMEM queue and wait for all of it;
//   * rounding points are the unfused path's: GEMM output (+ bias, GELU in fp32) -> fp16, residual add in fp16, LayerNorm =
//     exact two-pass statistics on the fp16 row, normalised value -> fp16 (rowgemm.hip's prologue).
// The launch is M / 32 blocks (256 at cfg-2 level 0: one per CU), each ingesting the block's 2.9 MB of weights from its XCD's
// L2 (they fit: 4 MB); the bound is that ingest, not the matrix cores.
// GroupNorm statistics of `out` for up to two consumer GroupNorms leave as fixed-point atomics like every other producer's.
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct RowChainArgs {
    const h16 *a, *res1, *res2;
    h16 *out;
    const h16 *w0, *w1, *w2, *w3;      // to_out | GEGLU projection (LN folded, value / gate interleaved) | FF2 | proj_out: fragment-packed
    const float *b0, *b1, *b2, *b3;
    unsigned long long *gn1, *gn2;
    int M, lda, ldr1, ldr2, ldo;
    int gnT, gnG, cpg1, choff1, cpg2, choff2;
    float eps;
};

#ifndef RC_RD
#define RC_RD 12       // weight ring depth in k steps (1 KB per step, wave and tile); 8 / 12 / 16: the same on warm weights, 12 best on cold ones and in the frame (7.744 -> 7.710 ms, tools/jobs/r6q.sh)
#endif
// analysis builds only (tools/variant_libs.sh build rowchain RC_X "..."; results are NOT the layer chain): 1 GELU -> identity, 2 no ring
// refills inside a k loop (the first RD steps' fragments are reused), 4 no MFMAs, 8 no activation fragment reads inside a k loop,
// 16 no ring at all (one fragment pair loaded once per kernel)
#ifndef RC_X
#define RC_X 0
#endif
// 32-row weight tiles per wave: 2 = five waves at C = 320 (one SIMD hosts two of them), 1 = ten waves (3, 3, 2, 2 per SIMD)
#ifndef RC_TAIL_NT
#define RC_TAIL_NT 1
#endif
#ifndef RC_HEAD_NT
#define RC_HEAD_NT 1
#endif
namespace {
constexpr int RC_BM = 32;

// sum over the 8 adjacent lanes that hold one activation row (as rowgemm.hip)
__device__ __forceinline__ float rc_row_sum8(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    return v;
}

// element offset (halfs) of channel c (a multiple of 4) of token `tok` in a tile of the activation layout
__device__ __forceinline__ int rc_addr(int c, int tok) {
    const int q = c >> 3;
    return ((q * RC_BM) + (tok ^ (2 * (q & 7)))) * 8 + (c & 7);
}

// the first RD k steps of NT consecutive 32-row weight tiles (SK k steps each) of this wave
template <int SK, int RD, int NT>
__device__ __forceinline__ void rc_ring_request(h16x8 (&wr)[RD][NT], const h16 *wp, int wlane) {
    long long ro = 0;                                       // running wave-uniform offset, opaque to the optimiser (rowgemm.hip)
#pragma unroll
    for (int s = 0; s < RD; ++s) {
        asm volatile("" : "+s"(ro));
#pragma unroll
        for (int i = 0; i < NT; ++i) if (!(RC_X & 16)) wr[s][i] = l2d_ld8(wp + (ro + (long long)i * SK * 512) + wlane);
        ro += 512;
    }
    __builtin_amdgcn_sched_barrier(0);
}

// acc[i] = W tile i (32 rows x 16 SK) . tile^T (16 SK x 32 tokens): straight-line, the ring refilled RD k steps ahead
template <int SK, int RD, int NT>
__device__ __forceinline__ void rc_kloop(f32x16 (&acc)[NT], h16x8 (&wr)[RD][NT], const h16 *wp, int wlane, const h16 *tile, const int (&xoff)[4]) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    h16x8 xf[2];
    long long wcur = (long long)RD * 512;
    asm volatile("" : "+s"(wcur));
    xf[0] = l2d_ld8(tile + xoff[0]);
#pragma unroll
    for (int s = 0; s < SK; ++s) {
        if (s + 1 < SK && !(RC_X & 8)) xf[(s + 1) & 1] = l2d_ld8(tile + ((s + 1) >> 2) * (64 * RC_BM) + xoff[(s + 1) & 3]);
        else if (s + 1 < SK) xf[(s + 1) & 1] = xf[s & 1];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (!(RC_X & 4)) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[s % RD][i], xf[s & 1], acc[i], 0, 0, 0);
            else acc[i][s & 15] += (float)wr[s % RD][i][0] * (float)xf[s & 1][0];
        }
        if (s + RD < SK && !(RC_X & 18)) {
#pragma unroll
            for (int i = 0; i < NT; ++i) wr[s % RD][i] = l2d_ld8(wp + (wcur + (long long)i * SK * 512) + wlane);
            wcur += 512;
            asm volatile("" : "+s"(wcur));
        }
        __builtin_amdgcn_sched_barrier(0);                  // the refills stay HERE: RD - 1 k steps ahead of their use
    }
}
}  // namespace

template <int C, int NT>
__global__ __launch_bounds__(C * 2 / NT) void rowchain_tail_kernel(RowChainArgs a) {
    constexpr int BM = RC_BM, NW = C / (32 * NT), SK = C / 16, H4 = 4 * C, SK2 = H4 / 16, RD = RC_RD, NTHR = NW * 64;
    constexpr int SLOTS = C / 8, RSTEP = NTHR / SLOTS, LPT = BM / RSTEP;        // 16-byte slots per row; rows per load pass; passes
    constexpr int NPASS = (2 * H4) / (NW * NT * 32);                            // GEGLU passes of C packed rows
    static_assert(C % 64 == 0 && NTHR % SLOTS == 0 && BM % RSTEP == 0 && (2 * H4) % (NW * NT * 32) == 0 && SK >= RD, "geometry");
    extern __shared__ __attribute__((aligned(16))) h16 smem[];                  // the ONLY LDS object
    h16 *X = smem, *Hh = smem + BM * C, *Gt = smem + 2 * BM * C;
    float *bl = reinterpret_cast<float *>(smem + 2 * BM * C + BM * H4);         // b0 [C] | b1 [2 H4] | b2 [C] | b3 [C]
    float *bl0 = bl, *bl1 = bl + C, *bl2 = bl + C + 2 * H4, *bl3 = bl + 2 * C + 2 * H4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int wlane = lane * 8;
    int xoff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xoff[u] = (((2 * u + lh) * BM) + (l32 ^ (4 * u + 2 * lh))) * 8;

    // ---- the attention output a -> X, the residual stream h1 -> H: 16 bytes per lane, a row's 40 slots by 40 adjacent threads
    const int slot = tid % SLOTS, r0 = tid / SLOTS;
    h16x8 wr[RD][NT];
    if (RC_X & 16) {
#pragma unroll
        for (int s = 0; s < RD; ++s)
#pragma unroll
            for (int i = 0; i < NT; ++i) wr[s][i] = l2d_ld8(a.w0 + (s * NT + i) * 512 + wlane);
    }
    {
        h16x8 va[LPT], vr[LPT];
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const long long row = m0 + r0 + RSTEP * i;
            va[i] = l2d_ld8(a.a + row * a.lda + slot * 8);
            vr[i] = l2d_ld8(a.res1 + row * a.ldr1 + slot * 8);
        }
        // the biases of all four layers -> LDS (C == threads: one element of b0 / b2 / b3 and 8 of b1 per thread), requested in
        // front of the weight ring so that their LDS stores do not wait for it
        static_assert(NTHR >= C, "bias staging: one thread per channel (the first C threads)");
        const int bt = tid < C ? tid : 0;
        const float vb0 = a.b0[bt], vb2 = a.b2[bt], vb3 = a.b3[bt];
        float vb1[2 * H4 / C];
#pragma unroll
        for (int k = 0; k < 2 * H4 / C; ++k) vb1[k] = a.b1[bt + C * k];
        __builtin_amdgcn_sched_barrier(0);
        rc_ring_request<SK, RD, NT>(wr, a.w0 + (long long)(wave * NT) * SK * 512, wlane);       // (behind the rows: rowgemm.hip)
        if (tid < C) {
            bl0[tid] = vb0; bl2[tid] = vb2; bl3[tid] = vb3;
#pragma unroll
            for (int k = 0; k < 2 * H4 / C; ++k) bl1[tid + C * k] = vb1[k];
        }
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int row = r0 + RSTEP * i;
            const int dst = ((slot * BM) + (row ^ (2 * (slot & 7)))) * 8;
            l2d_st8(X + dst, va[i]);
            l2d_st8(Hh + dst, vr[i]);
        }
    }
    __syncthreads();

    f32x16 acc[NT];
    // ---- stage 1: h2 = to_out(a) + b + h1 -> H
    rc_kloop<SK, RD, NT>(acc, wr, a.w0 + (long long)(wave * NT) * SK * 512, wlane, X, xoff);
    rc_ring_request<SK, RD, NT>(wr, a.w1 + (long long)(wave * NT) * SK * 512, wlane);           // GEGLU pass 0
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int c = (wave * NT + i) * 32 + 8 * g4 + 4 * lh;
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(bl0 + c);
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[i][4 * g4 + e] + bb[e]);
            h16x4 *hp = reinterpret_cast<h16x4 *>(Hh + rc_addr(c, l32));
            *hp = o + *hp;
        }
    __syncthreads();
    // ---- LayerNorm(h2) -> X: exact two-pass statistics, 8 lanes per row (gamma / beta live in the packed GEGLU weights)
    if (tid < BM * 8) {
        const int r = tid >> 3, j = tid & 7;
        constexpr int NS8 = SLOTS / 8;
        h16x8 v[NS8];
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NS8; ++i) {
            v[i] = l2d_ld8(Hh + (((j + 8 * i) * BM) + (r ^ (2 * j))) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) t += (float)v[i][e];
        }
        const float mean = rc_row_sum8(t) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NS8; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - mean; q += d * d; }
        const float rstd = rsqrtf(rc_row_sum8(q) / (float)C + a.eps);
#pragma unroll
        for (int i = 0; i < NS8; ++i) {
            h16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)v[i][e] - mean) * rstd);
            l2d_st8(X + (((j + 8 * i) * BM) + (r ^ (2 * j))) * 8, o);
        }
    }
    __syncthreads();
    // ---- stage 2: G = GEGLU(LN(h2)): NPASS passes of C packed rows (8 value rows, their 8 gate rows, ...: ops.rowgemm_geglu_perm)
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int t0 = p * NW * NT + wave * NT;
        rc_kloop<SK, RD, NT>(acc, wr, a.w1 + (long long)t0 * SK * 512, wlane, X, xoff);
        if (p + 1 < NPASS) rc_ring_request<SK, RD, NT>(wr, a.w1 + (long long)(t0 + NW * NT) * SK * 512, wlane);
        else rc_ring_request<SK2, RD, NT>(wr, a.w2 + (long long)(wave * NT) * SK2 * 512, wlane);         // FF2
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const float *bp = bl1 + (t0 + i) * 32 + 4 * lh + 16 * g2;
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bp), bg = *reinterpret_cast<const f32x4 *>(bp + 8);
                h16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (h16)((acc[i][8 * g2 + e] + bv[e]) * ((RC_X & 1) ? acc[i][8 * g2 + 4 + e] + bg[e] : l2d_gelu(acc[i][8 * g2 + 4 + e] + bg[e])));
                *reinterpret_cast<h16x4 *>(Gt + rc_addr((t0 + i) * 16 + 8 * g2 + 4 * lh, l32)) = o;
            }
    }
    __syncthreads();
    // ---- stage 3: h3 = FF2(G) + b + h2 -> X
    rc_kloop<SK2, RD, NT>(acc, wr, a.w2 + (long long)(wave * NT) * SK2 * 512, wlane, Gt, xoff);
    rc_ring_request<SK, RD, NT>(wr, a.w3 + (long long)(wave * NT) * SK * 512, wlane);           // proj_out
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int c = (wave * NT + i) * 32 + 8 * g4 + 4 * lh;
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(bl2 + c);
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[i][4 * g4 + e] + bb[e]);
            const int ad = rc_addr(c, l32);
            *reinterpret_cast<h16x4 *>(X + ad) = o + *reinterpret_cast<const h16x4 *>(Hh + ad);
        }
    __syncthreads();
    // ---- stage 4: out = proj_out(h3) + b + x (the block's input, from HBM: requested in front of the k loop)
    h16x8 vx[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) vx[i] = l2d_ld8(a.res2 + (long long)(m0 + r0 + RSTEP * i) * a.ldr2 + slot * 8);
    __builtin_amdgcn_sched_barrier(0);
    rc_kloop<SK, RD, NT>(acc, wr, a.w3 + (long long)(wave * NT) * SK * 512, wlane, X, xoff);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int c = (wave * NT + i) * 32 + 8 * g4 + 4 * lh;
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(bl3 + c);
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[i][4 * g4 + e] + bb[e]);
            *reinterpret_cast<h16x4 *>(Hh + rc_addr(c, l32)) = o;                               // (H is dead: staging for the row phase)
        }
    __syncthreads();
    // whole rows, 16 bytes per lane; a thread's slot (8 channels) is the same for all its rows: per-channel GroupNorm sums in registers
    const bool gn = a.gn1 != nullptr;
    float gs[4], gq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
    const h16x2 ones2 = {(h16)1.0f, (h16)1.0f};
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int row = r0 + RSTEP * i;
        const h16x8 v = l2d_ld8(Hh + ((slot * BM) + (row ^ (2 * (slot & 7)))) * 8) + vx[i];
        l2d_st8(a.out + (long long)(m0 + row) * a.ldo + slot * 8, v);
        if (gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const h16x2 pr = {v[2 * e], v[2 * e + 1]};
                gs[e] = __builtin_amdgcn_fdot2(pr, ones2, gs[e], false);
                gq[e] = __builtin_amdgcn_fdot2(pr, pr, gq[e], false);
            }
        }
    }
    if (gn) {
        // statistics of what was just stored, per channel pair, reduced to the consumers' groups inside the block (as rowgemm.hip)
        float *red = reinterpret_cast<float *>(Gt);                                              // [threads][8]: G is dead
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = gs[e]; red[tid * 8 + 4 + e] = gq[e]; }
        __syncthreads();
        float *chs1 = red + NTHR * 8, *chs2 = chs1 + C / 2;
        if (tid < C / 2) {
            const int c8 = tid >> 2, e = tid & 3;
            float s = 0.f, q = 0.f;
            for (int r = 0; r < RSTEP; ++r) { s += red[(r * SLOTS + c8) * 8 + e]; q += red[(r * SLOTS + c8) * 8 + 4 + e]; }
            chs1[tid] = s; chs2[tid] = q;
        }
        __syncthreads();
        const int bsmp = m0 / a.gnT;
        l2d_gn_flush(a.gn1, a.gnG, a.cpg1 >> 1, a.choff1 >> 1, bsmp, chs1, chs2, 0, C / 2, tid);
        l2d_gn_flush(a.gn2, a.gnG, a.cpg2 >> 1, a.choff2 >> 1, bsmp, chs1, chs2, 0, C / 2, tid);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Head segments of a transformer block, the same machinery: TWO dependent layers in one launch --
//     h   = A( norm?(x) ) + bA (+ resA)      A = proj_in behind the block's GroupNorm (statistics from x's producers: the prologue of
//                                            rowgemm.hip) or an attention's to_out with its residual      -> stored (the residual stream)
//     out = B( LayerNorm(h) ) + bB           B = q | k | v of the next attention (NBP passes of C packed rows; TRL: the LAST pass leaves
//                                            transposed, V^T[sample][channel][token] for the flash kernel) or the cross-attention's to_q
// (reference: attention.py:102-110,221-250; motion_module.py:273-279,401-427).  Replaces two row-GEMM launches and the round trip of h.
struct RowHeadArgs {
    const h16 *x, *resA;
    h16 *hout, *out, *outT;
    const h16 *wA, *wB;
    const float *bA, *bB;              // bB may be null (q | k | v carry no bias)
    const long long *gnacc;            // GroupNorm prologue: [samples][G][2] fixed-point statistics of x, or null
    long long sT;
    int M, ldx, ldrA, ldh, ldo, ldt, T, G;
    float eps_gn, eps_ln;
};

template <int C, int NBP, bool TRL, int NT>
__global__ __launch_bounds__(C * 2 / NT) void rowchain_head_kernel(RowHeadArgs a) {
    constexpr int BM = RC_BM, NW = C / (32 * NT), SK = C / 16, RD = RC_RD, NTHR = NW * 64;
    constexpr int SLOTS = C / 8, RSTEP = NTHR / SLOTS, LPT = BM / RSTEP, NS8 = SLOTS / 8;
    static_assert(C % 64 == 0 && NTHR % SLOTS == 0 && BM % RSTEP == 0 && NTHR >= C && SK >= RD, "geometry");
    extern __shared__ __attribute__((aligned(16))) h16 smem[];                  // the ONLY LDS object
    h16 *X = smem, *Hh = smem + BM * C, *S0 = smem + 2 * BM * C;                // S0 | S1: staging of B's output tiles (S1: + padding for V^T)
    constexpr int STG = C * (BM + 8) > BM * C ? C * (BM + 8) : BM * C;          // halfs per staging tile (channel-major V^T rows are BM + 8 wide)
    h16 *S1 = S0 + STG;
    float *bl = reinterpret_cast<float *>(S1 + STG);                            // bA [C] | bB [NBP C] | GroupNorm: sc [C] | sh [C] | rstd [32] | shift [32]
    float *blA = bl, *blB = bl + C, *tab = bl + C + NBP * C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int wlane = lane * 8;
    int xoff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xoff[u] = (((2 * u + lh) * BM) + (l32 ^ (4 * u + 2 * lh))) * 8;
    const int slot = tid % SLOTS, r0 = tid / SLOTS;
    const bool gnp = a.gnacc != nullptr, resa = a.resA != nullptr;

    h16x8 wr[RD][NT];
    {
        h16x8 va[LPT], vr[LPT];
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const long long row = m0 + r0 + RSTEP * i;
            va[i] = l2d_ld8(a.x + row * a.ldx + slot * 8);
            vr[i] = resa ? l2d_ld8(a.resA + row * a.ldrA + slot * 8) : l2d_zero8();
        }
        const int bt = tid < C ? tid : 0;                      // (the first C threads stage the biases: one channel each)
        const float vbA = a.bA[bt];
        float vbB[NBP];
#pragma unroll
        for (int k = 0; k < NBP; ++k) vbB[k] = a.bB ? a.bB[bt + C * k] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        rc_ring_request<SK, RD, NT>(wr, a.wA + (long long)(wave * NT) * SK * 512, wlane);
        if (tid < C) {
            blA[tid] = vbA;
#pragma unroll
            for (int k = 0; k < NBP; ++k) blB[tid + C * k] = vbB[k];
        }
        if (gnp) {
            // (rstd, -mean rstd) per channel of this block's sample, exactly as rowgemm.hip's prologue 2 (gamma / beta live in wA / bA)
            float *rstd_s = tab + 2 * C, *shf_s = rstd_s + 32;
            if (tid < a.G) {
                const long long *src = a.gnacc + ((long long)(m0 / a.T) * a.G + tid) * 2;
                const float s1 = (float)((double)src[0] * (1.0 / 1048576.0));
                const float q1 = (float)((double)src[1] * (1.0 / 4096.0));
                const float inv = 1.0f / ((float)a.T * (float)(C / a.G));
                const float mean = s1 * inv;
                const float var = fmaxf(q1 * inv - mean * mean, 0.f);
                const float rstd = rsqrtf(var + a.eps_gn);
                rstd_s[tid] = rstd;
                shf_s[tid] = -mean * rstd;
            }
            __syncthreads();
            if (tid < C) {
                const int g = (int)(((float)tid + 0.5f) * ((float)a.G / (float)C));
                tab[tid] = rstd_s[g];
                tab[C + tid] = shf_s[g];
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int row = r0 + RSTEP * i;
            const int dst = ((slot * BM) + (row ^ (2 * (slot & 7)))) * 8;
            h16x8 v = va[i];
            if (gnp) {
                const int c0 = slot * 8;
                const f32x4 sa = *reinterpret_cast<const f32x4 *>(tab + c0), sb = *reinterpret_cast<const f32x4 *>(tab + c0 + 4);
                const f32x4 ha = *reinterpret_cast<const f32x4 *>(tab + C + c0), hb = *reinterpret_cast<const f32x4 *>(tab + C + c0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (h16)((float)va[i][e] * sa[e] + ha[e]);
                    v[4 + e] = (h16)((float)va[i][4 + e] * sb[e] + hb[e]);
                }
            }
            l2d_st8(X + dst, v);
            l2d_st8(Hh + dst, vr[i]);
        }
    }
    __syncthreads();

    f32x16 acc[NT];
    // ---- layer A: h = A(x) + bA (+ resA) -> H
    rc_kloop<SK, RD, NT>(acc, wr, a.wA + (long long)(wave * NT) * SK * 512, wlane, X, xoff);
    rc_ring_request<SK, RD, NT>(wr, a.wB + (long long)(wave * NT) * SK * 512, wlane);           // B, pass 0
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int c = (wave * NT + i) * 32 + 8 * g4 + 4 * lh;
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(blA + c);
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[i][4 * g4 + e] + bb[e]);
            h16x4 *hp = reinterpret_cast<h16x4 *>(Hh + rc_addr(c, l32));
            *hp = resa ? o + *hp : o;
        }
    __syncthreads();
    // ---- h -> HBM (whole rows: the 8 lanes of a row store 128 contiguous bytes per step) and LayerNorm(h) -> X
    if (tid < BM * 8) {
        const int r = tid >> 3, j = tid & 7;
        h16x8 v[NS8];
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NS8; ++i) {
            v[i] = l2d_ld8(Hh + (((j + 8 * i) * BM) + (r ^ (2 * j))) * 8);
            l2d_st8(a.hout + (long long)(m0 + r) * a.ldh + (j + 8 * i) * 8, v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) t += (float)v[i][e];
        }
        const float mean = rc_row_sum8(t) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NS8; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - mean; q += d * d; }
        const float rstd = rsqrtf(rc_row_sum8(q) / (float)C + a.eps_ln);
#pragma unroll
        for (int i = 0; i < NS8; ++i) {
            h16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)v[i][e] - mean) * rstd);
            l2d_st8(X + (((j + 8 * i) * BM) + (r ^ (2 * j))) * 8, o);
        }
    }
    __syncthreads();
    // ---- layer B: NBP passes of C packed rows; every pass stages its tile (double-buffered) and leaves as whole rows / V^T rows
#pragma unroll
    for (int p = 0; p < NBP; ++p) {
        const int t0 = p * NW * NT + wave * NT;
        rc_kloop<SK, RD, NT>(acc, wr, a.wB + (long long)t0 * SK * 512, wlane, X, xoff);
        if (p + 1 < NBP) rc_ring_request<SK, RD, NT>(wr, a.wB + (long long)(t0 + NW * NT) * SK * 512, wlane);
        h16 *St = (p & 1) ? S1 : S0;
        constexpr int PT = BM + 8;
        const bool tr = TRL && p == NBP - 1;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c = (wave * NT + i) * 32 + 8 * g4 + 4 * lh;
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(blB + p * C + c);
                if (tr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) St[(c + e) * PT + l32] = (h16)(acc[i][4 * g4 + e] + bb[e]);      // channel-major
                } else {
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[i][4 * g4 + e] + bb[e]);
                    *reinterpret_cast<h16x4 *>(St + rc_addr(c, l32)) = o;
                }
            }
        __syncthreads();
        if (tr) {
            // V^T[sample][channel][token]: 16-byte pieces of 8 tokens (T % 32 == 0: the block lies in one sample)
            const int b = m0 / a.T, tb = m0 - b * a.T;
            h16 *ob = a.outT + (long long)b * a.sT + tb;
            constexpr int CPT = BM / 8;
            for (int idx = tid; idx < C * CPT; idx += NTHR) {
                const int ch = idx / CPT, cq = idx - ch * CPT;
                l2d_st8(ob + (long long)ch * a.ldt + cq * 8, l2d_ld8(St + ch * PT + cq * 8));
            }
        } else {
#pragma unroll
            for (int i = 0; i < LPT; ++i) {
                const int row = r0 + RSTEP * i;
                l2d_st8(a.out + (long long)(m0 + row) * a.ldo + p * C + slot * 8, l2d_ld8(St + ((slot * BM) + (row ^ (2 * (slot & 7)))) * 8));
            }
        }
    }
}

template <int C, int NBP, bool TRL>
static int launch_rh(const RowHeadArgs &a, hipStream_t s) {
    constexpr size_t STG = (size_t)(C * (RC_BM + 8) > RC_BM * C ? C * (RC_BM + 8) : RC_BM * C);
    constexpr size_t LDS = (size_t)(2 * RC_BM * C + 2 * STG) * 2 + (size_t)(C + NBP * C + 2 * C + 64) * 4;
    static_assert(LDS <= 163840, "tiles do not fit the CU's LDS");
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (LDS > 65536 && !attr_done) {
        if (hipFuncSetAttribute((const void *)rowchain_head_kernel<C, NBP, TRL, RC_HEAD_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) == hipSuccess) attr_done = true;
        else {
            l2d_set_error("rowchain head: the device refused %zu bytes of dynamic LDS (hipFuncSetAttribute: %s)", LDS, hipGetErrorString(hipGetLastError()));
            return L2D_ELAUNCH;                           // (nothing was launched)
        }
    }
    hipLaunchKernelGGL((rowchain_head_kernel<C, NBP, TRL, RC_HEAD_NT>), dim3(a.M / RC_BM), dim3(C * 2 / RC_HEAD_NT), LDS, s, a);
    return L2D_OK;
}

static int l2d_launch_rowchain_head(const l2d_op *op, hipStream_t s) {
    RowHeadArgs a;
    a.x = (const h16 *)op->p[0]; a.resA = (const h16 *)op->p[1]; a.hout = (h16 *)op->p[2]; a.out = (h16 *)op->p[3];
    a.wA = (const h16 *)op->p[4]; a.bA = (const float *)op->p[5]; a.wB = (const h16 *)op->p[6]; a.bB = (const float *)op->p[7];
    a.outT = (h16 *)op->p[8]; a.gnacc = (const long long *)op->p[14];
    a.M = op->i[0];
    const int C = op->i[1];
    a.ldx = op->i[2]; a.ldrA = op->i[3]; a.ldh = op->i[4]; a.ldo = op->i[5];
    const int nbp = op->i[7], trl = op->i[8];
    a.ldt = op->i[9]; a.T = op->i[10]; a.G = op->i[11];
    a.sT = op->l[0];
    a.eps_ln = op->f[0]; a.eps_gn = op->f[1];
    unsigned long long ptrs = (unsigned long long)a.x | (unsigned long long)a.resA | (unsigned long long)a.hout | (unsigned long long)a.out |
                              (unsigned long long)a.wA | (unsigned long long)a.wB | (unsigned long long)a.outT | (unsigned long long)a.bA | (unsigned long long)a.bB;
    const bool geo = (nbp == 1 && !trl) || (nbp == 3);
    if (!a.x || !a.hout || !a.wA || !a.bA || !a.wB || C != 320 || a.M <= 0 || (a.M % RC_BM) || !geo || a.ldx < C || a.ldh < C ||
        ((a.ldx | a.ldh) % 8) || (a.resA && (a.ldrA < C || (a.ldrA % 8))) || (ptrs & 15) || !(a.eps_ln > 0.f) ||
        (nbp - (trl ? 1 : 0) > 0 && (!a.out || a.ldo < (nbp - (trl ? 1 : 0)) * C || (a.ldo % 8))) ||
        (trl && (!a.outT || a.T <= 0 || (a.T % RC_BM) || (a.M % a.T) || (a.ldt % 8) || a.ldt < a.T)) ||
        (a.gnacc && (a.T <= 0 || (a.T % RC_BM) || (a.M % a.T) || a.G <= 0 || a.G > 32 || (C % a.G) || !(a.eps_gn > 0.f)))) {
        l2d_set_error("rowchain head(tag %d): invalid arguments (M=%d C=%d passes=%d transposed=%d ldx=%d ldh=%d ldo=%d T=%d G=%d)", op->tag, a.M,
                      C, nbp, trl, a.ldx, a.ldh, a.ldo, a.T, a.G);
        return L2D_EINVAL;
    }
    // (round-5 advisor finding) blocks read x / resA up front while other blocks already write hout / out / outT: an in-place call
    // through the C ABI would corrupt rows silently; element offsets are formed in 64 bits but bounded like the tail's
    {
        const long long maxld = a.ldx > a.ldh ? a.ldx : a.ldh;
        const void *ins[2] = {a.x, a.resA}, *outs[3] = {a.hout, a.out, a.outT};
        bool alias = false;
        for (const void *i_ : ins)
            for (const void *o_ : outs) alias |= (i_ && o_ && i_ == o_);
        if (alias || (long long)a.M * maxld >= (1ll << 40) || (long long)a.M * (a.ldo > 0 ? a.ldo : 1) >= (1ll << 40)) {
            l2d_set_error("rowchain head(tag %d): outputs must not alias x / resA, M * ld must stay below 2^40 (M=%d)", op->tag, a.M);
            return L2D_EINVAL;
        }
    }
    L2D_DRY_RETURN();
    const int rc = nbp == 1 ? launch_rh<320, 1, false>(a, s) : (trl ? launch_rh<320, 3, true>(a, s) : launch_rh<320, 3, false>(a, s));
    if (rc != L2D_OK) return rc;
    return l2d_check_launch("rowchain head", op->tag);
}

template <int C>
static void launch_rc(const RowChainArgs &a, hipStream_t s) {
    constexpr size_t LDS = (size_t)(2 * RC_BM * C + RC_BM * 4 * C) * 2 + (size_t)(3 * C + 8 * C) * 4;
    static_assert(LDS <= 163840, "tiles do not fit the CU's LDS");
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (LDS > 65536 && !attr_done) {
        if (hipFuncSetAttribute((const void *)rowchain_tail_kernel<C, RC_TAIL_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) == hipSuccess) attr_done = true;
        else (void)hipGetLastError();
    }
    hipLaunchKernelGGL((rowchain_tail_kernel<C, RC_TAIL_NT>), dim3(a.M / RC_BM), dim3(C * 2 / RC_TAIL_NT), LDS, s, a);
}

int l2d_launch_rowchain(const l2d_op *op, hipStream_t s) {
    if (op->i[6] == 1) return l2d_launch_rowchain_head(op, s);       // i6: 0 = block tail, 1 = head segment (two layers)
    RowChainArgs a;
    a.a = (const h16 *)op->p[0]; a.res1 = (const h16 *)op->p[1]; a.res2 = (const h16 *)op->p[2]; a.out = (h16 *)op->p[3];
    a.w0 = (const h16 *)op->p[4]; a.b0 = (const float *)op->p[5]; a.w1 = (const h16 *)op->p[6]; a.b1 = (const float *)op->p[7];
    a.w2 = (const h16 *)op->p[8]; a.gn1 = (unsigned long long *)op->p[9]; a.gn2 = (unsigned long long *)op->p[10];
    a.b2 = (const float *)op->p[11]; a.w3 = (const h16 *)op->p[12]; a.b3 = (const float *)op->p[13];
    a.M = op->i[0];
    const int C = op->i[1];
    a.lda = op->i[2]; a.ldr1 = op->i[3]; a.ldr2 = op->i[4]; a.ldo = op->i[5];
    a.gnT = op->i[24]; a.gnG = op->i[25]; a.cpg1 = op->i[26]; a.choff1 = op->i[27]; a.cpg2 = op->i[28]; a.choff2 = op->i[29];
    a.eps = op->f[0];
    if (!a.gn1 && a.gn2) { a.gn1 = a.gn2; a.cpg1 = a.cpg2; a.choff1 = a.choff2; a.gn2 = nullptr; }
    unsigned long long ptrs = 0;
    for (int k = 0; k < 14; ++k) if (k != 9 && k != 10) ptrs |= (unsigned long long)op->p[k];
    if (!a.a || !a.res1 || !a.res2 || !a.out || !a.w0 || !a.w1 || !a.w2 || !a.w3 || !a.b0 || !a.b1 || !a.b2 || !a.b3 ||
        C != 320 || a.M <= 0 || (a.M % RC_BM) || a.lda < C || a.ldr1 < C || a.ldr2 < C || a.ldo < C ||
        ((a.lda | a.ldr1 | a.ldr2 | a.ldo) % 8) || (ptrs & 15) || !(a.eps > 0.f) || (long long)a.M * a.lda >= (1ll << 40)) {
        l2d_set_error("rowchain(tag %d): invalid arguments (M=%d C=%d lda=%d ldr1=%d ldr2=%d ldo=%d): C = 320, M %% 32 == 0, 16-byte aligned "
                      "operands, all four packed layers and their fp32 biases", op->tag, a.M, C, a.lda, a.ldr1, a.ldr2, a.ldo);
        return L2D_EINVAL;
    }
    if (a.gn1) {
        if (a.gnT <= 0 || (a.gnT % RC_BM) || (a.M % a.gnT) || a.gnG <= 0 || a.gnG > 32 || a.cpg1 <= 0 || (a.gn2 && a.cpg2 <= 0) ||
            ((a.cpg1 | a.choff1) & 1) || (a.gn2 && ((a.cpg2 | a.choff2) & 1))) {
            l2d_set_error("rowchain(tag %d): GroupNorm statistics need T %% 32 == 0 (T=%d), even group sizes and offsets", op->tag, a.gnT);
            return L2D_EINVAL;
        }
    }
    L2D_DRY_RETURN();
    launch_rc<320>(a, s);
    return l2d_check_launch("rowchain", op->tag);
}
