// Token-row GEMM for the transformer linear layers of the UNet on gfx950: out = epi( norm(x) . W^T ), K <= 2048.
//
// Replaces, per launch, what the reference runs as two or three modules in a row:
//   nn.LayerNorm -> nn.Linear            (attention.py:182-205 norm1/2/3 + to_q/k/v / ff; motion_module.py:355-361 norms + qkv / ff)
//   GroupNorm    -> proj_in (Linear)     (attention.py:57,62,102-110; motion_module.py:181-182,273-279)
//   Linear (+ bias, + residual, GEGLU)   (attention.py:89,173-204; motion_module.py:207,360)
// and the q | k | v projections of the spatial self-attention as ONE launch that also writes V already transposed.
//
// Why a second GEMM kernel next to igemm.hip.  In the frame, 318 of the 378 GEMM launches are these linear layers; with the
// norms in front of them they took 5.5 of 9.3 ms (profiles/r3z_frame_trace.csv) at 70-350 TFLOP/s, because a 64x64 / 128x128
// tile kernel that stages BOTH operands through an LDS ring pays ~700 prologue + ~300 epilogue instructions and one block
// barrier per K step for 5-20 K steps of real work, and every LayerNorm / GroupNorm in front of it is one more launch on the
// dispatch floor.  This kernel has a different structure:
//   * a block owns BM = 32 (or 64) token rows and ALL of K: the activation tile is loaded ONCE (16-byte loads, 128-byte row
//     pieces), normalised on the way and parked in LDS as the MFMA B operand.  The affine part of the norm is NOT applied
//     here: gamma is folded into the packed weight (W diag(gamma)) and beta into the bias (b + W beta) at pack time -- exact
//     in real arithmetic; the fp16 rounding point moves from "norm output" to "normalised value" -- so the prologue needs
//     no parameter loads: LayerNorm = exact two-pass statistics per row (the row is re-read from LDS by the thread that
//     wrote it), GroupNorm = (mean, rstd) per group from the producers' fixed-point accumulators;
//   * weights never touch LDS: they are packed at load time in MFMA-fragment order ([n tile 32][k step 16][lane][8 halfs], so a
//     wave's A fragment is one contiguous, perfectly coalesced 1 KB load) and stream L2 -> VGPR through a register ring
//     (8-16 fragments in flight per wave), requested from the first instruction of the kernel, i.e. under the activation
//     load and the normalisation.  Waves own disjoint output channels, so there is nothing to share through LDS;
//   * the K loop has NO block barrier and no LDS-DMA bookkeeping: per k step one conflict-free ds_read_b128 of the
//     activation fragment (k-slab-major LDS image with an XOR on the token index; read one step ahead), NT x MT
//     v_mfma_f32_32x32x16_f16, NT refills.  For K = 320 / 640 / 1280 the loop is straight-line code, so hipcc's waitcnt
//     insertion counts every fragment exactly (`vmcnt(n)`, n = loads issued after the one being consumed); a
//     sched_barrier per k step keeps the scheduler from sinking the refills to just-in-time (which it does to save
//     registers: seen as `load; vmcnt(1); mfma` pairs in the ISA, i.e. a ring of depth 2);
//   * epilogue through LDS (the dead activation tile): bias / GEGLU in fp32 -> fp16 tile -> whole-row 16-byte stores with the
//     residual added on the way, GroupNorm statistics of the output for the NEXT GroupNorm accumulated like igemm.hip does;
//     the V part of a q|k|v launch is staged channel-major and leaves as V^T rows.
// Rounding points: normalised activations -> fp16, GEMM output (+ bias, activation in fp32) -> fp16, residual add in fp16.
//
// Roofline note: with BM = 32 every weight fragment (1 KB) feeds ONE 32x32x16 MFMA (32 cycles on its SIMD), i.e.
// 128 B/clk/CU of L1 traffic at full matrix rate against a 64 B/clk/CU path: this structure tops out at ~0.5 of the MFMA
// peak (BM = 64: 1.0).  It is built for the latency-bound launches (1-15 GFLOP each), not for the 3x3 convolutions.
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct RowGemmArgs {
    const h16 *x, *w;
    const float *bias;
    const h16 *res;
    h16 *out, *outT;
    const long long *gnacc;        // prologue 2: [samples][G][2] fixed-point statistics of x (filled by x's producers)
    unsigned long long *gn1, *gn2; // GroupNorm statistics of the OUTPUT for up to two consumers (as igemm.hip)
    long long sT;                  // V^T output: elements between samples
    int M, K, Nout, ldx, ldo, ldr, ldt;
    int epi, pro, T, G, cpg;
    int nm, ny, ytr, tr_n0, order;
    int gnT, gnG, cpg1, choff1, cpg2, choff2;
    float eps, inv_cpg;
#ifdef L2D_PROBES
    unsigned long long *probe;     // analysis builds: 8 s_memtime stamps per block (thread 0), tools/rowgemm_probe.py
#endif
};

#ifdef L2D_PROBES
static unsigned long long *g_rowgemm_probe = nullptr;
extern "C" void l2d_rowgemm_set_probe(void *p) { g_rowgemm_probe = (unsigned long long *)p; }
#define RG_STAMP(i) do { if (a.probe && threadIdx.x == 0) a.probe[(unsigned long long)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define RG_STAMP(i) do { } while (0)
#endif

// sum over the 8 (or 16) adjacent lanes that hold one activation row, on DPP (no LDS crossbar round trips): xor 1 and xor 2
// inside the quad, then the mirror partner inside the half row (the other quad, whose lanes all hold that quad's sum), then
// the mirror partner inside the row of 16
__device__ __forceinline__ float rg_row_sum(float v, bool wide) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    if (wide) v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

template <int RD, int SK>
constexpr int rg_ring() {
    // K = 1280 rows are normalised through LDS (two passes) and carry more live state: a 24-deep ring spills there, 16 does not
    return SK > 0 ? ((SK >= 80 && RD > 16) ? 16 : (RD < SK ? RD : SK)) : 4;
}

// main loop + epilogue.  SK = number of 16-wide k steps when known at compile time (K = 320 / 640 / 1280), 0 = runtime K
// (any K % 64 == 0: groups of 4 k steps, double-buffered; the compiler drains the loads at the loop back-edge, i.e. one
// exposed L2 round trip per 4 k steps -- only the unit-test widths and the odd layer take this path).
template <int NT, int MT, int RD, int SK>
__device__ __forceinline__ void rowgemm_body(const RowGemmArgs &a, h16 *smem, h16x8 (&wr)[rg_ring<RD, SK>()][NT], const h16 *wp,
                                             int wlane, long long tstride, int S, int m0, int y, int NW, int wave, int lane,
                                             int tid, int nthr) {
    constexpr int BM = 32 * MT;
    const int l32 = lane & 31, lh = lane >> 5;
    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][mt][e] = 0.f;

    // activation fragment of k step s, token tile mt: 16-byte slot q = 2 s + lh of token 32 mt + l32, stored at
    // ((q * BM) + (token ^ 2 (q & 7))) * 16 bytes; with s = 4 s4 + u the swizzle term depends on u only
    int xoff[4][MT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xoff[u][mt] = (((2 * u + lh) * BM) + 32 * mt + (l32 ^ (4 * u + 2 * lh))) * 8;

    // ---- epilogue geometry: whole output rows, 16 bytes per lane: thread -> (row rr + it * RPP, 8-channel chunk cc); cc is the
    // same in every pass, which is what lets a thread keep per-channel GroupNorm sums in registers.  The residual rows are
    // requested HERE, in front of the k loop (they are older than every weight refill in the in-order VMEM queue and land under
    // the MFMAs), not after it where they would be one more exposed round trip.
    const int BNp = NW * NT * 32;                            // packed weight rows of this block
    const int BNo = a.epi == 1 ? BNp >> 1 : BNp;             // output columns
    const int nb_p = y * BNp, nb_o = y * BNo;
    constexpr int RPPC = 16 / NT;                            // rows per pass without GEGLU (64 NW threads / (4 NW NT) chunks)
    constexpr int EIT = (BM + RPPC - 1) / RPPC;
    const int CPR = BNo >> 3;
    const int RPP = nthr / CPR;
    const int rr = tid / CPR, cc = tid - rr * CPR;
    const bool on = rr < RPP;
    // (128-token tiles keep 64-128 accumulator registers: their residual rows are fetched in the epilogue instead)
    constexpr bool RES_PRE = MT <= 2;
    h16x8 resv[RES_PRE ? EIT : 1];
    if (RES_PRE && a.res && y < a.ytr) {
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            int row = rr + it * RPP;
            row = (on && row < BM && m0 + row < a.M) ? row : 0;    // (clamped: an unconditional load, selected below)
            resv[it] = l2d_ld8(a.res + (long long)(m0 + row) * a.ldr + nb_o + (on ? cc : 0) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (SK > 0) {
        constexpr int RDC = rg_ring<RD, SK>();               // (the first RDC k steps were requested at kernel entry)
        h16x8 xf[2][MT];
        long long wcur = RDC * 512;                          // element offset of the next fragment of tile 0 to request
        asm volatile("" : "+s"(wcur));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[0][mt] = l2d_ld8(smem + xoff[0][mt]);
#pragma unroll
        for (int s = 0; s < SK; ++s) {
            if (s + 1 < SK) {                                // next step's activation fragment: its LDS latency hides under this step's MFMAs
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[(s + 1) & 1][mt] = l2d_ld8(smem + ((s + 1) >> 2) * (64 * BM) + xoff[(s + 1) & 3][mt]);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[i][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[s % RDC][i], xf[s & 1][mt], acc[i][mt], 0, 0, 0);
            if (s + RDC < SK) {
#pragma unroll
                for (int i = 0; i < NT; ++i) wr[s % RDC][i] = l2d_ld8(wp + (wcur + i * tstride) + wlane);
                // one running (wave-uniform) offset, opaque to the optimiser: otherwise the address of every refill of the
                // straight-line loop is computed at the top and stays live -- 64-bit VGPR pairs by the dozen, spills at K = 1280
                wcur += 512;
                asm volatile("" : "+s"(wcur));
            }
            __builtin_amdgcn_sched_barrier(0);               // the refills stay HERE: RDC - 1 k steps ahead of their use
        }
    } else {
        h16x8 wb[4][NT];
        auto group = [&](h16x8 (&cur)[4][NT], h16x8 (&nxt)[4][NT], int s0) {
            if (s0 + 4 < S) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < NT; ++i) nxt[u][i] = l2d_ld8(wp + (i * tstride + (long long)(s0 + 4 + u) * 512) + wlane);
            }
            const h16 *xb = smem + (s0 >> 2) * (64 * BM);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                h16x8 xf[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[mt] = l2d_ld8(xb + xoff[u][mt]);
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[i][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[u][i], xf[mt], acc[i][mt], 0, 0, 0);
            }
        };
        for (int s0 = 0; s0 < S; s0 += 8) {
            group(wr, wb, s0);
            if (s0 + 4 < S) group(wb, wr, s0 + 4);
        }
    }

    // ------------------------------------------------------------------------------------------------- epilogue
    RG_STAMP(4);                                             // k loop done (issue side)
    h16 *os = smem;
    __syncthreads();                                         // every wave is done reading the activation tile
    if (y >= a.ytr) {
        // transposed part (V^T[sample][channel][token] for the flash kernel): the tile is staged channel-major -- a lane holds
        // ONE token and 16 channels, so consecutive lanes write consecutive tokens of a channel row (2-byte LDS stores) -- and
        // leaves as 16-byte pieces of 8 tokens.  (T % BM == 0: the block lies in one sample, every row is valid.)
        const int pt = BM + 8;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int tp = (wave * NT + i) * 32;
            const float *bp = a.bias ? a.bias + nb_p + tp + 4 * lh : nullptr;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                f32x4 bb = {0.f, 0.f, 0.f, 0.f};
                if (bp) bb = *reinterpret_cast<const f32x4 *>(bp + 8 * g4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        os[(tp + 8 * g4 + 4 * lh + e) * pt + 32 * mt + l32] = (h16)(acc[i][mt][4 * g4 + e] + bb[e]);
            }
        }
        __syncthreads();
        const int b = m0 / a.T, tb = m0 - b * a.T;
        h16 *ob = a.outT + (long long)b * a.sT + (long long)(nb_p - a.tr_n0) * a.ldt + tb;
        constexpr int CPT = BM / 8;                           // 16-byte chunks per channel row
        for (int idx = tid; idx < BNp * CPT; idx += nthr) {
            const int ch = idx / CPT, c = idx - ch * CPT;
            l2d_st8(ob + (long long)ch * a.ldt + c * 8, l2d_ld8(os + ch * pt + c * 8));
        }
        return;
    }
    const int pitch = BNo + 8;                               // halfs; row stride = 16 B mod 32 B
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tp = (wave * NT + i) * 32;                 // block-local packed column of the tile
        const float *bp = a.bias ? a.bias + nb_p + tp + 4 * lh : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            h16 *orow = os + (32 * mt + l32) * pitch;
            if (a.epi == 1) {
                // GEGLU: packed tile rows [0,8) value, [8,16) gate of channels c..c+7, [16,24) / [24,32) of c+8..c+15:
                // register groups (0,1) and (2,3) hold value / gate of the SAME channels in the same lane
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bp + 16 * g2);
                    const f32x4 bg = *reinterpret_cast<const f32x4 *>(bp + 16 * g2 + 8);
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[i][mt][8 * g2 + e] + bv[e];
                        const float g = acc[i][mt][8 * g2 + 4 + e] + bg[e];
                        o[e] = (h16)(v * l2d_gelu(g));
                    }
                    *reinterpret_cast<h16x4 *>(orow + (tp >> 1) + 8 * g2 + 4 * lh) = o;
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 bb = {0.f, 0.f, 0.f, 0.f};
                    if (bp) bb = *reinterpret_cast<const f32x4 *>(bp + 8 * g4);
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[i][mt][4 * g4 + e] + bb[e]);
                    *reinterpret_cast<h16x4 *>(orow + tp + 8 * g4 + 4 * lh) = o;
                }
            }
        }
    }
    // whole rows, 16 bytes per lane (the residual rows were requested before the k loop)
    __syncthreads();
    RG_STAMP(5);                                             // tile staged in LDS, residual requested
    const bool gn = a.gn1 != nullptr;
    float gs[4], gq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
    const h16x2 ones2 = {(h16)1.0f, (h16)1.0f};
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
        const int row = rr + it * RPP;
        if (!on || row >= BM || m0 + row >= a.M) continue;
        h16x8 v = l2d_ld8(os + row * pitch + cc * 8);
        if (a.res) {
            if constexpr (RES_PRE) v = v + resv[it];
            else v = v + l2d_ld8(a.res + (long long)(m0 + row) * a.ldr + nb_o + cc * 8);
        }
        l2d_st8(a.out + (long long)(m0 + row) * a.ldo + nb_o + cc * 8, v);
        if (gn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const h16x2 pr = {v[2 * e], v[2 * e + 1]};
                gs[e] = __builtin_amdgcn_fdot2(pr, ones2, gs[e], false);
                gq[e] = __builtin_amdgcn_fdot2(pr, pr, gq[e], false);
            }
        }
    }
    RG_STAMP(6);                                             // row stores issued
    if (gn) {
        // statistics of what was just stored (the fp16 values the consumer GroupNorm will read), per channel pair,
        // reduced to the consumer's groups inside the block, two integer atomics per (consumer, overlapped group)
        float *red = reinterpret_cast<float *>(smem + BM * pitch);              // [threads][8], behind the staged tile
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = gs[e]; red[tid * 8 + 4 + e] = gq[e]; }
        __syncthreads();
        float *chs1 = red + nthr * 8, *chs2 = chs1 + (BNo >> 1);
        if (tid < (BNo >> 1)) {
            const int c8 = tid >> 2, e = tid & 3;
            float s = 0.f, q = 0.f;
            for (int r = 0; r < RPP; ++r) { s += red[(r * CPR + c8) * 8 + e]; q += red[(r * CPR + c8) * 8 + 4 + e]; }
            chs1[tid] = s; chs2[tid] = q;
        }
        __syncthreads();
        const int bsmp = m0 / a.gnT;
        l2d_gn_flush(a.gn1, a.gnG, a.cpg1 >> 1, a.choff1 >> 1, bsmp, chs1, chs2, nb_o >> 1, BNo >> 1, tid);
        l2d_gn_flush(a.gn2, a.gnG, a.cpg2 >> 1, a.choff2 >> 1, bsmp, chs1, chs2, nb_o >> 1, BNo >> 1, tid);
    }
}

template <int NT, int MT, int RD, int SK, int MAXT>
__global__ __launch_bounds__(MAXT) void rowgemm_kernel(RowGemmArgs a) {
    constexpr int BM = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) h16 smem[];     // the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthr = blockDim.x, NW = nthr >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RG_STAMP(0);                                              // block entry

    // XCD-aware block order (blocks are dispatched round-robin over the 8 XCDs): every XCD runs a contiguous range of work
    // items, token-tile major (activations dominate: level 0) or weight-band major (order 1: the weight band of an XCD stays
    // in ITS L2, the small activation matrix enters all eight)
    const int nwg = a.nm * a.ny;
    int wgid;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int mtile, y;
    if (a.order) { y = wgid / a.nm; mtile = wgid - y * a.nm; }
    else { mtile = wgid / a.ny; y = wgid - mtile * a.ny; }
    const int m0 = mtile * BM;
    const int S = SK > 0 ? SK : (a.K >> 4);                   // k steps of 16
    const int KK = SK > 0 ? SK * 16 : a.K;
    const int t0 = (y * NW + wave) * NT;                      // this wave's first 32-row weight tile

    // ---- weight stream: fragment-packed weights of this wave's tiles.  The first ring of fragments is requested right behind
    // the FIRST group of activation loads (below), not in front of them: VMEM returns in order and a CU ingests only ~20-30
    // bytes per clock when every block of the launch bursts at once, so 16 KB of weights per wave queued ahead of the
    // activation rows delayed the normalisation -- the head of the block's critical path -- by the whole ring's transfer time.
    // (wave-uniform base in SGPRs + one constant per-lane offset: the straight-line k loop then needs no per-load address VGPRs)
    const h16 *wp = a.w + (long long)t0 * S * 512;
    const int wlane = lane * 8;
    const long long tstride = (long long)S * 512;             // halfs between consecutive 32-row weight tiles
    constexpr int RING = rg_ring<RD, SK>();
    h16x8 wr[RING][NT];
    auto request_ring = [&]() {
        long long ro = 0;                                     // running wave-uniform element offset, opaque to the optimiser (the
#pragma unroll                                                // OFFSET, not the pointer: a pointer that went through an asm
        for (int s = 0; s < RING; ++s) {                      // statement loses its address space and its loads become flat_load,
            asm volatile("" : "+s"(ro));                      // which count on lgkmcnt too and cannot be waited for selectively)
#pragma unroll
            for (int i = 0; i < NT; ++i) wr[s][i] = l2d_ld8(wp + (ro + i * tstride) + wlane);
            ro += 512;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    RG_STAMP(1);                                              // weight ring requested

    // ---- GroupNorm prologue: (rstd, -mean rstd) per channel of this block's sample (the block lies inside one sample; gamma
    // and beta live in the packed weight / bias).  One dependent read of the 512-byte accumulator block, no parameter loads.
    float *tab = reinterpret_cast<float *>(smem + BM * KK + 8 * BM);   // sc[K] | sh[K] | rstd[32] | shift[32]  (behind the tile + slack)
    if (a.pro == 2) {
        float *rstd_s = tab + 2 * KK, *shf_s = rstd_s + 32;
        if (tid < a.G) {
            const long long *src = a.gnacc + ((long long)(m0 / a.T) * a.G + tid) * 2;
            const float s = (float)((double)src[0] * (1.0 / 1048576.0));
            const float q = (float)((double)src[1] * (1.0 / 4096.0));
            const float inv = 1.0f / ((float)a.T * (float)a.cpg);
            const float mean = s * inv;
            const float var = fmaxf(q * inv - mean * mean, 0.f);
            const float rstd = rsqrtf(var + a.eps);
            rstd_s[tid] = rstd;
            shf_s[tid] = -mean * rstd;
        }
        __syncthreads();
        for (int c = tid; c < KK; c += nthr) {
            const int g = (int)(((float)c + 0.5f) * a.inv_cpg);
            tab[c] = rstd_s[g];
            tab[KK + c] = shf_s[g];
        }
        __syncthreads();
    }

    RG_STAMP(2);                                              // GroupNorm tables ready (prologue 2)
    // ---- activation tile -> LDS.  Thread (row r = tid / LPR, j = tid % LPR) moves the 16-byte slots q = j + LPR i of its
    // row (LPR = 8 lanes per row, 16 when the block has the threads and K % 128 == 0: half as many dependent load groups for
    // K = 1280): the lanes of a row read 128 / 256 contiguous bytes per step; slot q of token r lands at
    // ((q * BM) + (r ^ 2 (q & 7))) * 16 B.  Loads go out in groups of 10 slots per thread (K <= 640: the whole row at once),
    // unconditionally (rows beyond M are clamped and zeroed afterwards: a select around a load would become a branch per load).
    // Order of requests: every (row, group) pair of this thread but the last is loaded and consumed in a loop; the LAST pair is
    // requested, then -- in straight-line code, so that the compiler can count the loads behind it -- the weight ring, then
    // the last pair is consumed: the activation rows never queue behind the weights, and the ring is still in flight
    // (not drained by a conservative vmcnt(0)) when the k loop starts.
    {
        constexpr int XG = 10;                                // slots per thread and load group: K <= 640 (1280 with 16 lanes per row) is ONE group
        const bool wide = nthr >= 16 * BM && (KK & 127) == 0;
        const int lsh = wide ? 4 : 3, LPR = 1 << lsh;
        const int KS = KK >> (3 + lsh);
        const int nch = (KS + XG - 1) / XG;                   // load groups per row
        const int j = tid & (LPR - 1);
        const int dstep = 8 * LPR * BM;                       // halfs between slots q and q + LPR
        const int r0 = tid >> lsh, rstep = nthr >> lsh;
        const bool has_row = r0 < BM;
        const int r_last = has_row ? r0 + ((BM - 1 - r0) / rstep) * rstep : 0;
        const bool ln_reg = a.pro == 1 && nch == 1;           // the thread's whole row slice is one load group: LayerNorm in registers

        auto load_group = [&](h16x8 (&v)[XG], int r, int i0) {
            const int m = m0 + r;
            const h16 *src = a.x + (long long)(m < a.M ? m : 0) * a.ldx + j * 8;
#pragma unroll
            for (int u = 0; u < XG; ++u) v[u] = l2d_ld8(src + (i0 + u < KS ? i0 + u : KS - 1) * (8 * LPR));
            __builtin_amdgcn_sched_barrier(0);
        };
        // consumes one group; `s` carries the row sum of the LDS-based LayerNorm across the groups of a row
        auto consume_group = [&](h16x8 (&v)[XG], int r, int i0, float &s) {
            const bool rv = m0 + r < a.M;
            h16 *dst = smem + (j * BM + (r ^ (2 * (j & 7)))) * 8;
            if (ln_reg) {
                // exact two-pass statistics on the registers, one LDS store per slot
                float t = 0.f;
#pragma unroll
                for (int u = 0; u < XG; ++u) {
                    if (u >= KS) break;
                    if (!rv) v[u] = l2d_zero8();
#pragma unroll
                    for (int e = 0; e < 8; ++e) t += (float)v[u][e];
                }
                const float mean = rg_row_sum(t, wide) / (float)KK;
                float q = 0.f;
#pragma unroll
                for (int u = 0; u < XG; ++u) {
                    if (u >= KS) break;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = (float)v[u][e] - mean; q += d * d; }
                }
                const float rstd = rsqrtf(rg_row_sum(q, wide) / (float)KK + a.eps);
#pragma unroll
                for (int u = 0; u < XG; ++u) {
                    if (u >= KS) break;
                    h16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)v[u][e] - mean) * rstd);
                    l2d_st8(dst + u * dstep, o);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < XG; ++u) {
                if (i0 + u >= KS) break;
                if (!rv) v[u] = l2d_zero8();
                if (a.pro == 2) {
                    const int c0 = (j + LPR * (i0 + u)) * 8;
                    const f32x4 sa = *reinterpret_cast<const f32x4 *>(tab + c0), sb = *reinterpret_cast<const f32x4 *>(tab + c0 + 4);
                    const f32x4 ha = *reinterpret_cast<const f32x4 *>(tab + KK + c0), hb = *reinterpret_cast<const f32x4 *>(tab + KK + c0 + 4);
                    h16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (h16)((float)v[u][e] * sa[e] + ha[e]);
                        o[4 + e] = (h16)((float)v[u][4 + e] * sb[e] + hb[e]);
                    }
                    v[u] = rv ? o : l2d_zero8();
                } else if (a.pro == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += (float)v[u][e];
                }
                l2d_st8(dst + (i0 + u) * dstep, v[u]);
            }
        };
        // LayerNorm of a row wider than one load group: exact two-pass on the raw row parked in LDS (re-read by its own thread)
        auto finish_row_lds = [&](int r, float s) {
            h16 *dst = smem + (j * BM + (r ^ (2 * (j & 7)))) * 8;
            const float mean = rg_row_sum(s, wide) / (float)KK;
            float q = 0.f;
#pragma unroll 5
            for (int i = 0; i < KS; ++i) {
                const h16x8 v = l2d_ld8(dst + i * dstep);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)v[e] - mean; q += d * d; }
            }
            const float rstd = rsqrtf(rg_row_sum(q, wide) / (float)KK + a.eps);
#pragma unroll 5
            for (int i = 0; i < KS; ++i) {
                const h16x8 v = l2d_ld8(dst + i * dstep);
                h16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)v[e] - mean) * rstd);
                l2d_st8(dst + i * dstep, o);
            }
        };

        float srow = 0.f;
        if (has_row) {
            for (int r = r0; r < BM; r += rstep) {
                const int ng = (r == r_last) ? nch - 1 : nch;  // the last group of the last row is peeled below
                for (int c = 0; c < ng; ++c) {
                    h16x8 v[XG];
                    load_group(v, r, c * XG);
                    consume_group(v, r, c * XG, srow);
                }
                if (r != r_last) {
                    if (a.pro == 1 && !ln_reg) finish_row_lds(r, srow);
                    srow = 0.f;
                }
            }
        }
        h16x8 vl[XG];
        if (has_row) load_group(vl, r_last, (nch - 1) * XG);
        request_ring();
        if (has_row) {
            consume_group(vl, r_last, (nch - 1) * XG, srow);
            if (a.pro == 1 && !ln_reg) finish_row_lds(r_last, srow);
        }
    }
    __syncthreads();
    RG_STAMP(3);                                              // activation tile normalised and visible
    rowgemm_body<NT, MT, RD, SK>(a, smem, wr, wp, wlane, tstride, S, m0, y, NW, wave, lane, tid, nthr);
}

template <int NT, int MT, int RD, int SK, int MAXT>
static void launch_rg(const RowGemmArgs &a, int nthr, size_t lds, hipStream_t s) {
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (lds > 65536 && !attr_done) {   // > 64 KB of dynamic LDS must be opted into once per kernel (not inside a capture:
        // the plan's first run is always direct)
        if (hipFuncSetAttribute((const void *)rowgemm_kernel<NT, MT, RD, SK, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) == hipSuccess)
            attr_done = true;
        else
            (void)hipGetLastError();
    }
    hipLaunchKernelGGL((rowgemm_kernel<NT, MT, RD, SK, MAXT>), dim3(a.nm * a.ny), dim3(nthr), lds, s, a);
}

// K = 320 / 640 / 1280 (the SD-1.5 widths: 20 / 40 / 80 k steps) get the straight-line loop, anything else the generic one
template <int NT, int MT, int RD, int MAXT>
static void launch_k(const RowGemmArgs &a, int nthr, size_t lds, hipStream_t s) {
    switch (a.K) {
        case 320: launch_rg<NT, MT, RD, 20, MAXT>(a, nthr, lds, s); break;
        case 640: launch_rg<NT, MT, RD, 40, MAXT>(a, nthr, lds, s); break;
        case 1280: launch_rg<NT, MT, RD, 80, MAXT>(a, nthr, lds, s); break;
        default: launch_rg<NT, MT, RD, 0, MAXT>(a, nthr, lds, s); break;
    }
}

int l2d_launch_rowgemm(const l2d_op *op, hipStream_t s) {
    RowGemmArgs a;
    a.x = (const h16 *)op->p[0]; a.w = (const h16 *)op->p[1]; a.bias = (const float *)op->p[2]; a.res = (const h16 *)op->p[3];
    a.out = (h16 *)op->p[4]; a.gnacc = (const long long *)op->p[7];
    a.outT = (h16 *)op->p[8];
    a.gn1 = (unsigned long long *)op->p[9]; a.gn2 = (unsigned long long *)op->p[10];
    a.M = op->i[0]; a.K = op->i[1]; a.Nout = op->i[2]; a.ldx = op->i[3]; a.ldo = op->i[4]; a.ldr = op->i[5];
    a.epi = op->i[6]; a.pro = op->i[7]; a.T = op->i[9]; a.G = op->i[10];
    const int NW = op->i[12], NT = op->i[13], MT = op->i[14];
    const int ntr = op->i[15];                      // number of TRAILING 32-row weight tiles whose output is stored transposed
    a.ldt = op->i[16]; a.order = op->i[17] ? 1 : 0;
    a.sT = op->l[0];
    a.eps = op->f[0];
#ifdef L2D_PROBES
    a.probe = g_rowgemm_probe;
#endif
    a.gnT = op->i[24]; a.gnG = op->i[25]; a.cpg1 = op->i[26]; a.choff1 = op->i[27]; a.cpg2 = op->i[28]; a.choff2 = op->i[29];
    if (!a.gn1 && a.gn2) { a.gn1 = a.gn2; a.cpg1 = a.cpg2; a.choff1 = a.choff2; a.gn2 = nullptr; }
    const int BM = 32 * MT;
    const int tiles = a.Nout > 0 ? a.Nout / 32 : 0;
    // (launch bounds of the instantiations below: NT <= 2 up to 8 waves, NT >= 3 up to 5)
    const bool geom_ok = NW >= 1 && NW <= (NT >= 3 ? 5 : 8) && NT >= 1 && NT <= 4 && (MT == 1 || MT == 2 || MT == 4) &&
                         !(MT >= 2 && NT > 2) && !(MT == 4 && a.K != 320) &&   // (128-token tiles: K = 320 only -- 80 KB of LDS)
                         tiles > 0 && (a.Nout % 32) == 0 && (tiles % (NW * NT)) == 0 && ntr >= 0 && ntr <= tiles &&
                         (ntr % (NW * NT)) == 0;
    if (!a.x || !a.w || a.M <= 0 || a.K <= 0 || (a.K % 64) || a.K > 2048 || !geom_ok || (a.ldx % 8) || a.ldx < a.K ||
        a.epi < 0 || a.epi > 1 || a.pro < 0 || a.pro > 2 ||
        (ntr < tiles && (!a.out || (a.ldo % 8))) || (a.res && (a.ldr % 8)) ||
        (a.epi == 1 && (!a.bias || a.res || ntr != 0)) ||
        (a.pro == 2 && (!a.gnacc || a.T <= 0 || (a.T % BM) || (a.M % a.T) || a.G <= 0 || a.G > 32 || (a.K % a.G))) ||
        (ntr > 0 && (!a.outT || a.T <= 0 || (a.T % BM) || (a.M % a.T) || (a.ldt % 8) || a.ldt < a.T)) ||
        (((unsigned long long)a.x | (unsigned long long)a.w | (unsigned long long)a.out | (unsigned long long)a.res |
          (unsigned long long)a.outT | (unsigned long long)a.bias) & 15)) {
        l2d_set_error("rowgemm(tag %d): invalid arguments (M=%d K=%d Nout=%d ldx=%d ldo=%d epi=%d pro=%d NW=%d NT=%d MT=%d ntr=%d T=%d)",
                      op->tag, a.M, a.K, a.Nout, a.ldx, a.ldo, a.epi, a.pro, NW, NT, MT, ntr, a.T);
        return L2D_EINVAL;
    }
    const int BNp = NW * NT * 32, BNo = a.epi == 1 ? BNp / 2 : BNp, nthr = 64 * NW;
    if (a.gn1) {
        if (a.gnT <= 0 || (a.gnT % BM) || (a.M % a.gnT) || a.gnG <= 0 || a.gnG > 32 || a.cpg1 <= 0 || (a.gn2 && a.cpg2 <= 0) ||
            ((a.cpg1 | a.choff1) & 1) || (a.gn2 && ((a.cpg2 | a.choff2) & 1)) || ntr != 0 || a.epi == 1 || nthr < 32) {
            l2d_set_error("rowgemm(tag %d): GroupNorm statistics need T %% %d == 0 (T=%d), even group sizes and offsets, no "
                          "transposed part", op->tag, BM, a.gnT);
            return L2D_EINVAL;
        }
    }
    a.cpg = a.pro == 2 ? a.K / a.G : 1;
    a.inv_cpg = 1.0f / (float)a.cpg;
    a.nm = (a.M + BM - 1) / BM;
    a.ny = tiles / (NW * NT);
    a.ytr = (tiles - ntr) / (NW * NT);
    a.tr_n0 = (tiles - ntr) * 32;
    // activation tile (+ one k step of slack: the loop reads one fragment ahead) + GroupNorm tables
    const size_t xs = (size_t)BM * a.K * 2 + (size_t)BM * 16 + (a.pro == 2 ? (size_t)(2 * a.K + 64) * 4 : 0);
    size_t os = (size_t)BM * (BNo + 8) * 2 + (a.gn1 ? (size_t)nthr * 32 + (size_t)BNo * 4 : 0);
    if (ntr > 0 && (size_t)BNp * (BM + 8) * 2 > os) os = (size_t)BNp * (BM + 8) * 2;       // channel-major staging of the V^T part
    const size_t lds = xs > os ? xs : os;
    if (lds > 163840 || (long long)a.nm * a.ny >= (1 << 24)) {
        l2d_set_error("rowgemm(tag %d): tile does not fit (LDS %zu bytes, %d x %d blocks)", op->tag, lds, a.nm, a.ny);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    // ring depth (fragments in flight per wave and tile): up to 24 KB per wave for NT = 1 (the K = 1280 levels are bound by the
    // latency of their weight stream: 80 KB per wave), 8 KB per tile otherwise
    if (MT == 1) {
        switch (NT) {
            case 1: launch_k<1, 1, 24, 512>(a, nthr, lds, s); break;
            case 2: launch_k<2, 1, 8, 512>(a, nthr, lds, s); break;
            case 3: launch_k<3, 1, 5, 320>(a, nthr, lds, s); break;
            default: launch_k<4, 1, 4, 320>(a, nthr, lds, s); break;
        }
    } else if (MT == 2) {
        if (NT == 1) launch_k<1, 2, 16, 512>(a, nthr, lds, s);
        else launch_k<2, 2, 8, 512>(a, nthr, lds, s);
    } else {
        if (NT == 1) launch_rg<1, 4, 8, 20, 512>(a, nthr, lds, s);
        else launch_rg<2, 4, 4, 20, 512>(a, nthr, lds, s);
    }
    return l2d_check_launch("rowgemm", op->tag);
}
