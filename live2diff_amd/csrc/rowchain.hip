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
//     projection) through a register ring of 8 k steps per 32-row tile; every wave owns 64 of the 320 output columns of a pass
//     (2 tiles), a 4C-wide layer is 8 passes.  The whole chain is straight-line code (C is a template parameter: 280 k steps), so
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
        for (int i = 0; i < NT; ++i) wr[s][i] = l2d_ld8(wp + (ro + (long long)i * SK * 512) + wlane);
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
        if (s + 1 < SK) xf[(s + 1) & 1] = l2d_ld8(tile + ((s + 1) >> 2) * (64 * RC_BM) + xoff[(s + 1) & 3]);
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[s % RD][i], xf[s & 1], acc[i], 0, 0, 0);
        if (s + RD < SK) {
#pragma unroll
            for (int i = 0; i < NT; ++i) wr[s % RD][i] = l2d_ld8(wp + (wcur + (long long)i * SK * 512) + wlane);
            wcur += 512;
            asm volatile("" : "+s"(wcur));
        }
        __builtin_amdgcn_sched_barrier(0);                  // the refills stay HERE: RD - 1 k steps ahead of their use
    }
}
}  // namespace

template <int C>
__global__ __launch_bounds__(C) void rowchain_tail_kernel(RowChainArgs a) {
    constexpr int BM = RC_BM, NW = C / 64, NT = 2, SK = C / 16, H4 = 4 * C, SK2 = H4 / 16, RD = 8, NTHR = NW * 64;
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
        static_assert(NTHR == C, "bias staging assumes one thread per channel");
        const float vb0 = a.b0[tid], vb2 = a.b2[tid], vb3 = a.b3[tid];
        float vb1[2 * H4 / C];
#pragma unroll
        for (int k = 0; k < 2 * H4 / C; ++k) vb1[k] = a.b1[tid + C * k];
        __builtin_amdgcn_sched_barrier(0);
        rc_ring_request<SK, RD, NT>(wr, a.w0 + (long long)(wave * NT) * SK * 512, wlane);       // (behind the rows: rowgemm.hip)
        bl0[tid] = vb0; bl2[tid] = vb2; bl3[tid] = vb3;
#pragma unroll
        for (int k = 0; k < 2 * H4 / C; ++k) bl1[tid + C * k] = vb1[k];
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
                for (int e = 0; e < 4; ++e) o[e] = (h16)((acc[i][8 * g2 + e] + bv[e]) * l2d_gelu(acc[i][8 * g2 + 4 + e] + bg[e]));
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

template <int C>
static void launch_rc(const RowChainArgs &a, hipStream_t s) {
    constexpr size_t LDS = (size_t)(2 * RC_BM * C + RC_BM * 4 * C) * 2 + (size_t)(3 * C + 8 * C) * 4;
    static_assert(LDS <= 163840, "tiles do not fit the CU's LDS");
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (LDS > 65536 && !attr_done) {
        if (hipFuncSetAttribute((const void *)rowchain_tail_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) == hipSuccess) attr_done = true;
        else (void)hipGetLastError();
    }
    hipLaunchKernelGGL((rowchain_tail_kernel<C>), dim3(a.M / RC_BM), dim3(C), LDS, s, a);
}

int l2d_launch_rowchain(const l2d_op *op, hipStream_t s) {
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
