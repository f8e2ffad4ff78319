// Weight-streaming GEMM for the small-token half of the UNet on gfx950 (levels 2 / 3 / mid at the SD resolutions: M = N*T <= ~1k
// tokens): out = epi( LN?(x) . W^T ), linear layers and 3x3 stride-1 convolutions alike.
//
// Replaces, per launch (reference call sites): the 1280-channel 3x3 convs of ResnetBlock3D (resnet.py:194,214 via InflatedConv3d
// :57-65), the GEGLU / FF-out projections (attention.py:204,258; motion_module.py:360), the q/k/v/out projections
// (attention.py:173-194; stream_motion_module.py:99-147) and the LayerNorms in front of them (attention.py:182-205).
//
// Why a third GEMM kernel.  At M <= 512 these launches are weight streams: 2.07 GB of the frame's 2.56 GB of weights pass through
// 162 launches whose arithmetic intensity (120-380 flop per weight byte) sits at or below the machine balance, and the two tile
// kernels served them at 0.5-0.9 TB/s: igemm.hip stages BOTH operands through LDS with 64 / 128-token tiles
// (every block re-ingests its activation tile per channel tile, ~14-22 B/clk/CU by LDS-DMA), rowgemm.hip keeps 32 tokens resident
// and therefore reads every weight byte M / 32 times.  Here:
//   * a block owns 128 tokens (ALL of them at the 8x8 level) x BN = 32 NW output channels x one K slice (grid = row tiles x
//     channel tiles x S slices): every weight byte is fetched by exactly one wave per row tile -- once from HBM at M = 128;
//   * weights never touch LDS: fragment-packed at load time ([n tile 32][k step 16][lane][8 halfs], ops.pack_wsgemm /
//     pack_wsgemm_conv3x3), each consumer wave streams its ONE 32-row tile HBM / L2 -> VGPR through a register ring of 8-16
//     fragments (1 KB, perfectly coalesced, optionally non-temporal when no second row tile will read them; inline-asm loads
//     with hand-counted vmcnt waits, see ws_gload) and feeds each fragment to FOUR v_mfma_f32_32x32x16_f16 (one per 32-token
//     tile);
//   * the 128-token activation operand is K-chunked: 64 channels (16 KB) per stage through a 4-stage LDS ring filled by dedicated
//     LOADER wave(s) with global_load_lds (16 B per lane, no VGPR staging; the XOR bank swizzle is applied to the SOURCE slot
//     because the DMA image is lane-linear), one s_barrier per stage; a 3x3 conv is the same loop with (tap, channel chunk)
//     stages whose source row is the token's neighbour pixel or the zero page (padding), incl. the two-pointer channel concat;
//   * LayerNorm in front of the layer costs nothing in the loop: gamma / beta are folded into the packed weights (W diag(gamma),
//     b + W beta) and the normalisation itself is applied to the ACCUMULATOR, out = rstd (acc - mean colsum(W')) + b', with the
//     row statistics summed by the LOADER waves from the bytes they moved (v_dot2 on a read-back of their own DMA'd 16 bytes per
//     lane) -- the consumers' loop is identical with and without the norm, and the activations enter the matrix cores raw (fp16
//     as stored), which is closer to exact than the reference's fp16-rounded norm output;
//   * split-K with the reduction fused as in igemm.hip: fp32 partial tiles (and partial row statistics) leave as write-through
//     (sc1) buffer stores into a tile-private slab, an arrival counter picks the last block, which sums the S slabs in the fixed
//     order 0..S-1 (bit-repeatable) and runs the epilogue; no fences (igemm.hip explains why);
//   * epilogue = rowgemm.hip's: fp32 bias (+ per-sample time-embedding bias) / GEGLU -> fp16 tile in LDS -> whole rows with 16 B
//     per lane, residual added in fp16 on the way, GroupNorm statistics of the output as fixed-point integer atomics (per
//     sample: a 128-token tile may span samples at the 8x8 level), V^T staging for the flash kernel.
// Rounding points: GEMM output (+ bias, activation in fp32) -> fp16, residual add in fp16 (as igemm / rowgemm).
// This file is compiled WITHOUT packed fp32 VALU instructions (csrc/Makefile; DESIGN.md 7.0 has the measurement that led to it).
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define L2D_GPTR(p) ((__attribute__((address_space(1))) const void *)(p))
#define L2D_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

struct WsArgs {
    const h16 *x1, *x2, *w, *zero;
    const float *bias, *colsum, *rowbias;
    const h16 *res;
    h16 *out, *outT;
    float *ws;                     // split-K slabs: [tile][S][128 * BNp + 256] floats
    unsigned int *cnt;             // split-K arrival counters, one per (channel tile, row tile); zero before and after
    unsigned long long *gn1, *gn2;
    long long sT;
    int M, Ktot, C1, C2, CinP, ldx1, ldx2, ldo, ldr, ldt, ldrb, rows_per_bias;
    int taps, H, W, T;
    float invT, invW;
    int epi, pro;
    int nm, ny, S, cps, crem, ncpt, ytr, tr_n0;
    int gnT, gnG, cpg1, choff1, cpg2, choff2;
    int stat_off;                  // byte offset of the row-statistics block in LDS
    float eps;
#ifdef L2D_PROBES
    unsigned long long *probe;     // analysis builds: 64 s_memtime stamps per block (loader 0: 0..31, consumer 0: 32..63), tools/wsgemm_stamps.py
#endif
};

#ifdef L2D_PROBES
static unsigned long long *g_wsgemm_probe = nullptr;
extern "C" void l2d_wsgemm_set_probe(void *p) { g_wsgemm_probe = (unsigned long long *)p; }
#define WS_STAMP(i) do { if (a.probe && lane == 0) a.probe[(unsigned long long)blockIdx.x * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

// q = m / d for 0 <= m < 2^22 with a host-computed float reciprocal (one estimate, one correction step)
__device__ __forceinline__ int ws_div(int m, int d, float inv, int &rem) {
    int q = (int)((float)m * inv);
    int r = m - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

// Weight fragment loads and their waits are inline asm: hipcc's own waitcnt insertion drains every load that is in flight across
// a loop back-edge (s_waitcnt vmcnt(0) in the loop header: one exposed HBM round trip per iteration), and the K slice of a block is
// a run-time trip count.  The compiler does not see these loads, so it inserts nothing; the counted wait below is tied ("+v") to
// the fragment registers it covers, which orders the MFMAs that read them behind it.  Loads return in order: when at most N
// younger ones are outstanding, this one has landed.  (tests/test_kernel_resources.py checks in the ISA that no instruction
// other than these touches a ring register between its load and its wait.)
template <bool NTW>
__device__ __forceinline__ void ws_gload(h16x8 &dst, unsigned voff, const h16 *sbase) {
    if constexpr (NTW) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase));
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
template <int N>
__device__ __forceinline__ void ws_gwait(h16x8 &a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void ws_gwait(h16x8 &a, h16x8 &b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
// The activation-fragment reads (LDS) are asm for the same reason: behind inline asm hipcc waits lgkmcnt(0) before every use of a
// double-buffered fragment, i.e. for the reads it has just issued for the NEXT step.  LDS returns in order: with at most N
// younger reads outstanding the four fragments of this step have landed.
template <unsigned OFF>
__device__ __forceinline__ void ws_lread(h16x8 &dst, unsigned addr) {
    static_assert(OFF < 65536u, "ds_read immediate offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void ws_lwait(h16x8 &a, h16x8 &b, h16x8 &c, h16x8 &d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ws_lwait1(h16x8 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
// end of the K slice: everything this wave requested has landed.  ALL ring registers are operands: the last ring of a slice
// holds clamped duplicate requests that nothing consumes, and a register the compiler considers dead would be handed to other
// values while the hardware can still write it.
__device__ __forceinline__ void ws_gdrain8(h16x8 &a, h16x8 &b, h16x8 &c, h16x8 &d, h16x8 &e, h16x8 &f, h16x8 &g, h16x8 &h) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}

// NT: 32-row weight tiles per consumer wave; RDS: depth of the weight register ring in stages (4 k steps each); NL: loader waves;
// NTW: non-temporal weight loads;
constexpr int WS_PSTR = 320;   // floats per epilogue-parameter row in LDS (bias | column sums | 4 time-embedding rows): >= the widest block (10 waves x 32 columns)
// MAXW: launch bound in waves (sets the register budget: <= 6 waves -> 256 VGPRs, 10 -> 168)
template <int NT, int RDS, int NL, bool NTW, int MAXW>
__global__ __launch_bounds__(64 * MAXW) void wsgemm_kernel(WsArgs a) {
    constexpr int BM = 128, MT = 4, NS = 4;
    constexpr int STG = BM * 64;                               // halfs per ring stage (128 tokens x 64 channels)
    constexpr int IPL = 16 / NL;                               // DMA instructions per loader wave and stage
    extern __shared__ __attribute__((aligned(16))) h16 smem[];     // the ONLY LDS object: ring | (epilogue: tile | reduction) | statistics
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = (int)(blockDim.x >> 6) - NL;                // consumer waves
    const int nthr = NW * 64;                                  // threads that take part in the epilogue's row phase

    // XCD-aware bijective block order: every XCD runs a contiguous range of work items; the row tiles that share a weight band
    // are neighbours (the band leaves HBM once and is served to the other row tiles by that XCD's L2)
    const int nwg = a.nm * a.ny * a.S;
    int wgid;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int rest = wgid / a.nm, mtile = wgid - rest * a.nm;
    const int y = rest / a.S, z = rest - y * a.S;
    const int m0 = mtile * BM;
    const int c0 = z * a.cps + (z < a.crem ? z : a.crem);      // this block's K slice: chunks [c0, c1) of 64 columns
    const int n = a.cps + (z < a.crem ? 1 : 0);

    float *stat = reinterpret_cast<float *>(reinterpret_cast<char *>(smem) + a.stat_off);      // [128][2]
    // epilogue parameters of this block's columns, DMA'd into LDS by loader wave 0 before anything else (bias | column sums | up to
    // four time-embedding rows): read as cold global loads at the START of the epilogue they were ~1 us of exposed latency per block
    float *par = stat + 2 * BM + 16;

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][mt][e] = 0.f;
    const int l32 = lane & 31, lh = lane >> 5;

    // row phase geometry of the epilogue (whole output rows, 16 bytes per lane): thread -> (row rr + k RPP, 8-channel chunk cc),
    // k < 8.  The residual rows are requested EARLY where registers allow (the <= 6-wave forms): in front of the weight ring --
    // they are older than every fragment in the in-order VMEM queue, so the first counted wait of the k loop covers them and they
    // cost no round trip of their own; the 10-wave forms (168-register budget) request them right behind the k loop.
    const int BNp = NW * NT * 32;                              // packed weight rows of this block
    const int BNo = a.epi == 1 ? BNp >> 1 : BNp;               // output columns
    const int nb_p = y * BNp, nb_o = y * BNo;
    const bool cons = wave < NW;
    const int CPR = BNo >> 3, RPP = nthr / CPR;
    const int rr = tid / CPR, cc = tid - rr * CPR;
    const bool on = cons && rr < RPP;
    constexpr bool RES_EARLY = NT == 1 && MAXW <= 6;            // (the other forms run at their register budget)
    constexpr int EITMAX = 8 * NT, PF = RES_EARLY ? 8 : (NT == 1 ? 2 : 1);
    h16x8 resp[PF];
    auto request_residual = [&]() {
        if (a.res && on && y < a.ytr) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                int row = rr + k * RPP;
                row = (row < BM && m0 + row < a.M) ? row : 0;      // (clamped: an unconditional load)
                resp[k] = l2d_ld8(a.res + (long long)(m0 + row) * a.ldr + nb_o + cc * 8);
            }
        }
    };

    if (wave >= NW) {
        // ------------------------------------------------------------------------------------------ loader wave(s)
        // DMA instruction i (0..15) of a stage moves tokens 8 i .. 8 i + 7, lane -> (token 8 i + lane / 8, LDS slot position
        // lane % 8); the stage image is [token][8 slots of 16 B]; position p of token r holds channel slot p ^ ((r >> 1) & 7),
        // which makes the consumers' ds_read_b128 fragment reads (32 consecutive tokens, one slot) conflict-free.
        const int l = wave - NW;
        const int sub = lane >> 3, qpos = lane & 7;
        if (l == 0) WS_STAMP(0);                                // loader entry
        if (l == 0) {
            const int bnp = NW * NT * 32, nbp = y * bnp;
            for (int c = lane; c < ((bnp + 63) & ~63); c += 64) {   // (whole wave-instructions; lanes beyond the block's columns re-read its last one)
                const int cl = c < bnp ? c : bnp - 1;
                const int cbase = c - lane;                     // wave-uniform LDS position of this instruction
                if (a.bias) __builtin_amdgcn_global_load_lds(L2D_GPTR(a.bias + nbp + cl), L2D_LPTR(par + cbase), 4, 0, 0);
                if (a.pro == 1) __builtin_amdgcn_global_load_lds(L2D_GPTR(a.colsum + nbp + cl), L2D_LPTR(par + WS_PSTR + cbase), 4, 0, 0);
                if (a.rowbias) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        int mm = m0 + 32 * mt;
                        if (mm >= a.M) mm = a.M - 1;
                        const float *rbp = a.rowbias + (long long)(mm / a.rows_per_bias) * a.ldrb;
                        __builtin_amdgcn_global_load_lds(L2D_GPTR(rbp + nbp + cl), L2D_LPTR(par + 2 * WS_PSTR + mt * WS_PSTR + cbase), 4, 0, 0);
                    }
                }
            }
        }
        // per DMA instruction of this lane: byte offset of its token's pixel row in x1 / x2 (+ its 16-byte slot), and for 3x3
        // convs the 9-bit mask of the taps whose neighbour pixel lies inside the image (bit 4 = the pixel itself; rows beyond M: 0).
        // The per-stage work is then: one wave-uniform offset (channel chunk + neighbour displacement), per instruction one mask
        // test, one select against the zero page and one 64-bit add -- the issue loop of this wave is what paces a stage.
        int off1[IPL], off2[IPL], vm[IPL];
        {
            const int m = m0 + 8 * l + sub;
            int yy = 0, xx = 0;
            if (a.taps == 9) {
                int p, b = ws_div(m, a.T, a.invT, p);
                (void)b;
                yy = ws_div(p, a.W, a.invW, xx);
            }
#pragma unroll
            for (int j = 0; j < IPL; ++j) {
                const int i = l + j * NL, mm = m0 + 8 * i + sub;
                const int q8 = (((((i & 1) << 2) | (sub >> 1)) ^ qpos)) << 3;
                off1[j] = (mm * a.ldx1 + q8) * 2;
                off2[j] = (mm * a.ldx2 + q8) * 2;
                int msk = 1 << 4;
                if (a.taps == 9) {
                    const int xm = (xx > 0 ? 1 : 0) | 2 | (xx < a.W - 1 ? 4 : 0);
                    msk = (yy > 0 ? xm : 0) | (xm << 3) | (yy < a.H - 1 ? xm << 6 : 0);
                    xx += 8 * NL;                               // next token of this lane: 8 NL pixels further (W >= 8: at most two row wraps)
                    if (xx >= a.W) { xx -= a.W; ++yy; }
                    if (xx >= a.W) { xx -= a.W; ++yy; }
                    if (yy >= a.H) yy -= a.H;
                }
                vm[j] = mm < a.M ? msk : 0;
            }
        }
        // Stages come in SEGMENTS: runs of consecutive channel chunks of one tap and one input tensor.  Inside a segment every
        // source address advances by 128 bytes per stage, so the issue loop is one 64-bit add + one DMA per instruction; padding /
        // ragged rows walk through the zero region (p7: at least 2 CinP + 256 zero bytes) the same way.  A segment starts with one
        // mask test + select + add per instruction.
        int tap = 4, cc = c0;                                   // (linear layers: the centre tap, dy = dx = 0)
        if (a.taps == 9) { tap = c0 / a.ncpt; cc = c0 - tap * a.ncpt; }
        const char *zp = reinterpret_cast<const char *>(a.zero);
        const char *cur[IPL];
        int seg_left = 0;
        const int nc1 = a.C1 >> 6;
        auto begin_segment = [&]() {
            const int t3 = (tap * 11) >> 5;                     // tap / 3 for tap < 9
            const int delta = (t3 - 1) * a.W + (tap - 3 * t3 - 1);      // neighbour displacement in pixels (0 for linear layers)
            const bool first = cc < nc1;
            const char *xb = reinterpret_cast<const char *>(first ? a.x1 : a.x2) +
                             ((long long)delta * (first ? a.ldx1 : a.ldx2) + (first ? cc : cc - nc1) * 64) * 2;     // wave-uniform
            const int bit = 1 << tap;
#pragma unroll
            for (int j = 0; j < IPL; ++j) cur[j] = (vm[j] & bit) ? xb + (first ? off1[j] : off2[j]) : zp;
            seg_left = first ? nc1 - cc : a.ncpt - cc;
        };
        auto issue_stage = [&](int slot) {
            if (seg_left == 0) begin_segment();
            h16 *dst = smem + slot * STG;
#pragma unroll
            for (int j = 0; j < IPL; ++j) {
                const int i = l + j * NL;
                __builtin_amdgcn_global_load_lds(L2D_GPTR(cur[j]), L2D_LPTR(dst + i * 512), 16, 0, 0);
                cur[j] += 128;
            }
            --seg_left;
            if (++cc == a.ncpt) { cc = 0; ++tap; }
        };
        // LayerNorm fold: the row statistics (sum x, sum x^2 over this block's K slice) are taken HERE, by the loader wave(s), from
        // the bytes they moved: once its share of a stage has landed the wave reads it back from LDS -- lane-linear, the very
        // 16 bytes per lane it requested -- and accumulates per DMA instruction (= per token of this lane) with v_dot2.  The
        // consumers' loop is the same with and without the norm, and any number of consumer waves will do.
        float sx[IPL], sq[IPL];
#pragma unroll
        for (int j = 0; j < IPL; ++j) { sx[j] = 0.f; sq[j] = 0.f; }
        const bool stats = a.pro == 1;
        auto stage_stats = [&](int slot) {
            const unsigned src = (unsigned)(size_t)L2D_LPTR(smem) + (slot * STG + lane * 8) * 2;
            const h16x2 ones2 = {(h16)1.0f, (h16)1.0f};
#pragma unroll
            for (int j0 = 0; j0 < IPL; j0 += 8) {               // batches of 8 reads: one LDS latency per batch
                h16x8 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(src + (l + (j0 + j) * NL) * 1024));
                ws_lwait<0>(v[0], v[1], v[2], v[3]);
                ws_lwait<0>(v[4], v[5], v[6], v[7]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const h16x2 pr = {v[j][2 * e], v[j][2 * e + 1]};
                        sx[j0 + j] = __builtin_amdgcn_fdot2(pr, ones2, sx[j0 + j], false);
                        sq[j0 + j] = __builtin_amdgcn_fdot2(pr, pr, sq[j0 + j], false);
                    }
                }
            }
        };
        const int npre = n < NS - 1 ? n : NS - 1;
        if (l == 0) WS_STAMP(1);                                // descriptors ready
        for (int s = 0; s < npre; ++s) issue_stage(s);
        if (l == 0) WS_STAMP(2);                                // first stages requested
        int s = 0;
        for (; s + (NS - 1) < n; ++s) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPL) : "memory");     // stage s has landed (this wave's share)
            if (l == 0 && s < 12) WS_STAMP(3 + 2 * s);          // stage s landed
            __builtin_amdgcn_s_barrier();      // ... every loader's has; the consumers are done with stage s - 1
            if (l == 0 && s < 12) WS_STAMP(4 + 2 * s);          // barrier s passed
            issue_stage((s + NS - 1) & (NS - 1));
            if (stats) stage_stats(s & (NS - 1));               // (this wave's own share of stage s: landed above; its slot is refilled
        }                                                       //  by this wave's NEXT issue, behind these reads)
        for (; s < n; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (l == 0 && s < 12) WS_STAMP(3 + 2 * s);
            __builtin_amdgcn_s_barrier();
            if (l == 0 && s < 12) WS_STAMP(4 + 2 * s);
            if (stats) stage_stats(s & (NS - 1));
        }
        __builtin_amdgcn_s_barrier();          // (the consumers' last stage ends like every other one: with the barrier of a "next" stage)
        if (stats) {
            // the 8 lanes of a token (its 8 channel slots) are adjacent: sum them on DPP, lane qpos == 0 publishes the token's sums
            auto dpp = [](float x, auto ctrl) {
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
            };
#pragma unroll
            for (int j = 0; j < IPL; ++j) {
                float tx = sx[j], tq = sq[j];
                tx += dpp(tx, std::integral_constant<int, 0xB1>{}); tq += dpp(tq, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
                tx += dpp(tx, std::integral_constant<int, 0x4E>{}); tq += dpp(tq, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
                tx += dpp(tx, std::integral_constant<int, 0x141>{}); tq += dpp(tq, std::integral_constant<int, 0x141>{});    // row_half_mirror
                if (qpos == 0) { const int r = 8 * (l + j * NL) + sub; stat[2 * r] = tx; stat[2 * r + 1] = tq; }
            }
        }
        if (l == 0) WS_STAMP(28);                               // loader done
    } else {
        // ------------------------------------------------------------------------------------------ consumer waves
        if (wave == 0) WS_STAMP(32);                            // consumer entry
        if constexpr (RES_EARLY) {
            request_residual();
            __builtin_amdgcn_sched_barrier(0);                  // (in FRONT of the weight ring)
        }
        const int t0 = (y * NW + wave) * NT;                    // this wave's first 32-row weight tile
        const int KST = a.Ktot >> 4;                            // k steps of the whole contraction
        const h16 *wp = a.w + (long long)t0 * KST * 512;          // wave-uniform (SGPR) base of this wave's first tile
        unsigned voff[NT];                                      // per-lane byte offset inside a fragment (+ the tile stride)
#pragma unroll
        for (int i = 0; i < NT; ++i) voff[i] = (unsigned)lane * 16u + (unsigned)i * ((unsigned)KST * 1024u);
        int ks = c0 * 4;
        const int ks_last = (c0 + n) * 4 - 1;
        h16x8 wr[4 * RDS][NT];
#pragma unroll
        for (int j = 0; j < 4 * RDS; ++j) {
            const int kk = ks + j < ks_last ? ks + j : ks_last;
#pragma unroll
            for (int i = 0; i < NT; ++i) ws_gload<NTW>(wr[j][i], voff[i], wp + (long long)kk * 512);
        }
        unsigned xoff[4];                                       // LDS byte address of this lane's fragment slot of k step u (stage 0, token tile 0)
        const unsigned lds0 = (unsigned)(size_t)L2D_LPTR(smem);
#pragma unroll
        for (int u = 0; u < 4; ++u) xoff[u] = lds0 + (l32 * 64 + ((((2 * u + lh) ^ ((l32 >> 1) & 7))) << 3)) * 2;
        h16x8 xf[2][MT];
        if (wave == 0) WS_STAMP(33);                            // weight ring requested
        __builtin_amdgcn_s_barrier();                           // stage 0 has landed
        if (wave == 0) WS_STAMP(34);
        ws_lread<0>(xf[0][0], xoff[0]); ws_lread<4096>(xf[0][1], xoff[0]); ws_lread<8192>(xf[0][2], xoff[0]); ws_lread<12288>(xf[0][3], xoff[0]);

        // one stage = 4 k steps on ring positions 4 p .. 4 p + 3; the NEXT stage's barrier is met inside this stage's
        // last k step, after the stage's last fragment read, and the next stage's first fragments are fetched under that step's MFMAs
        // The LDS slot of a stage is a compile-time constant where the loop structure allows it (RDS == NS: the slot offsets fold into
        // the ds_read immediates and the k loop holds no VALU instruction at all); with RDS == NS / 2 the slot pair alternates at
        // run time (SLOT < 0: the slot is `slot_rt`) and costs one v_add per k step.
        auto do_stage = [&](auto pc, auto refill_c, auto slot_c, int slot_rt) {
            constexpr int p = decltype(pc)::value;
            constexpr bool REFILL = decltype(refill_c)::value;
            constexpr int SLOT = decltype(slot_c)::value;
            constexpr unsigned sb = SLOT < 0 ? 0u : (unsigned)SLOT * (STG * 2);                    // byte offsets of this / the next
            constexpr unsigned nb = SLOT < 0 ? 0u : (unsigned)((SLOT + 1) & (NS - 1)) * (STG * 2); // stage (static part)
            const unsigned sbr = SLOT < 0 ? (unsigned)slot_rt * (STG * 2) : 0u;                    // (run-time part)
            const unsigned nbr = SLOT < 0 ? (unsigned)((slot_rt + 1) & (NS - 1)) * (STG * 2) : 0u;
            // weights of k step u landed -> its MFMAs -> refill of its ring slot
            auto compute = [&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int j = 4 * p + u;
                // this k step's weight fragments have landed when only the younger requests are outstanding: NT per later k step
                // of the ring (main loop: all 4 RDS - 1 of them; the last stages issue no refills and count down)
                if constexpr (NT == 1) {
                    if constexpr (REFILL) ws_gwait<4 * RDS - 1>(wr[j][0]);
                    else ws_gwait<(4 * RDS - 1 - j > 0 ? 4 * RDS - 1 - j : 0)>(wr[j][0]);
                } else {
                    if constexpr (REFILL) ws_gwait<2 * (4 * RDS - 1)>(wr[j][0], wr[j][1]);
                    else ws_gwait<(4 * RDS - 1 - j > 0 ? 2 * (4 * RDS - 1 - j) : 0)>(wr[j][0], wr[j][1]);
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[i][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[j][i], xf[u & 1][mt], acc[i][mt], 0, 0, 0);
                if constexpr (REFILL) {
                    const int kn = ks + 4 * RDS;
                    const int kk = kn < ks_last ? kn : ks_last;   // (clamped: the last ring of the slice re-requests its final fragment)
#pragma unroll
                    for (int i = 0; i < NT; ++i) ws_gload<NTW>(wr[j][i], voff[i], wp + (long long)kk * 512);
                }
                ++ks;
                __builtin_amdgcn_sched_barrier(0);              // the refills stay HERE: 4 RDS - 1 k steps ahead of their use
            };
            auto step = [&](auto uc) {                          // k steps 0..2: request the next step's fragments, wait for this step's
                constexpr int u = decltype(uc)::value;
                const unsigned ad = xoff[u + 1] + sbr;           // (+ a static slot, in the immediates: <= 61440)
                ws_lread<sb>(xf[(u + 1) & 1][0], ad); ws_lread<sb + 4096>(xf[(u + 1) & 1][1], ad);
                ws_lread<sb + 8192>(xf[(u + 1) & 1][2], ad); ws_lread<sb + 12288>(xf[(u + 1) & 1][3], ad);
                ws_lwait<4>(xf[u & 1][0], xf[u & 1][1], xf[u & 1][2], xf[u & 1][3]);      // (LDS returns in order: only the 4 reads above are younger)
                __builtin_amdgcn_sched_barrier(0);              // (the next step's fragment reads stay in FRONT of this step's MFMAs)
                compute(uc);
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            // Last k step.  NOTHING asm-loaded from LDS may be in flight at a basic-block boundary: at control-flow merges the compiler
            // is free to COPY a fragment register it believes to be valid -- seen in the ISA as v_mov of a fragment whose ds_read had
            // not been waited for, and on the GPU as wrong rows whenever another kernel on the CU slowed the LDS down.  A stage is
            // therefore straight-line code that ends with every LDS read it issued waited for; there is no "is there a next stage"
            // branch: the LAST stage of a slice meets the loader's final barrier and fetches four fragments of a slot nobody will use
            // (~100 cycles once per block).  (The weight ring crosses the loop back-edge by design; tests/test_kernel_resources.py
            // replays every path of the ISA for both queues.)
            ws_lwait<0>(xf[1][0], xf[1][1], xf[1][2], xf[1][3]);
            __builtin_amdgcn_s_barrier();                       // this wave is done with the stage's slot; the next stage has landed
            {
                const unsigned ad = xoff[0] + nbr;
                ws_lread<nb>(xf[0][0], ad); ws_lread<nb + 4096>(xf[0][1], ad); ws_lread<nb + 8192>(xf[0][2], ad); ws_lread<nb + 12288>(xf[0][3], ad);
            }
            __builtin_amdgcn_sched_barrier(0);                  // (the next stage's first fragments are fetched under this step's MFMAs)
            compute(std::integral_constant<int, 3>{});
            ws_lwait<0>(xf[0][0], xf[0][1], xf[0][2], xf[0][3]);
#ifdef L2D_PROBES
            { const int sdone = (ks >> 2) - c0 - 1; if (wave == 0 && sdone < 12) WS_STAMP(35 + 2 * sdone); }   // stage done (issue side)
#endif
        };
        // main loop: RDS stages per iteration (static ring positions)
        using T_ = std::true_type;
        using F_ = std::false_type;
#define WS_C(v) std::integral_constant<int, (v)>{}
        // the last 1 .. RDS stages: their fragments are already in the ring (no refills).  Nested, so that the control-flow graph has
        // no path that skips a stage and runs a later one.
        // BASE >= 0: static slots BASE, BASE + 1, ...; BASE < 0: run-time slots s0, s0 + 1, ...
        auto tail = [&](auto base_c, int s0, int rem) {
            constexpr int B0 = decltype(base_c)::value;
#define WS_SL(k) WS_C(B0 < 0 ? -1 : ((B0 + (k)) & (NS - 1))), ((s0 + (k)) & (NS - 1))
            do_stage(WS_C(0), F_{}, WS_SL(0));
            if constexpr (RDS > 1) {
                if (rem > 1) {
                    do_stage(WS_C(1), F_{}, WS_SL(1));
                    if constexpr (RDS > 2) {
                        if (rem > 2) {
                            do_stage(WS_C(2), F_{}, WS_SL(2));
                            if constexpr (RDS > 3) {
                                if (rem > 3) do_stage(WS_C(3), F_{}, WS_SL(3));
                            }
                        }
                    }
                }
            }
#undef WS_SL
        };
        static_assert(RDS == NS || 2 * RDS == NS, "the k loop is written for a register ring of NS or NS / 2 stages");
        int s = 0;
        if constexpr (RDS == NS) {
            for (; s + RDS < n; s += RDS) {
                do_stage(WS_C(0), T_{}, WS_C(0), 0);
                do_stage(WS_C(1), T_{}, WS_C(1), 0);
                do_stage(WS_C(2), T_{}, WS_C(2), 0);
                do_stage(WS_C(3), T_{}, WS_C(3), 0);
            }
            tail(WS_C(0), 0, n - s);
        } else {
            for (; s + RDS < n; s += RDS) {
                do_stage(WS_C(0), T_{}, WS_C(-1), s & (NS - 1));
                do_stage(WS_C(1), T_{}, WS_C(-1), (s + 1) & (NS - 1));
            }
            tail(WS_C(-1), s, n - s);
        }
        // (the clamped duplicates of the last ring: nothing of this wave's is in flight beyond here)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 4 * RDS; j += 8)
                ws_gdrain8(wr[j][i], wr[j + 1][i], wr[j + 2][i], wr[j + 3][i], wr[j + 4][i], wr[j + 5][i], wr[j + 6][i], wr[j + 7][i]);
#undef WS_C
        if (wave == 0) WS_STAMP(60);                            // k loop done (issue side)
    }

    // ------------------------------------------------------------------------------------------------- epilogue
    if constexpr (!RES_EARLY) request_residual();
    __syncthreads();                                           // ring idle; the slice's row statistics are in LDS
    if (wave == 0) WS_STAMP(61);                               // all waves done with the loop

    if (a.S > 1) {
        // split-K, reduction fused (protocol of igemm.hip): partial accumulators (and partial row statistics) leave as
        // write-through stores into the tile's slab, lane-linear; the block that arrives last sums the S slabs in order 0..S-1
        constexpr int AUX_SC1 = 16;
        const int SLABF = BM * BNp + 2 * BM;                   // floats per (tile, slice)
        const int tile = y * a.nm + mtile;
        float *slab = a.ws + (long long)tile * a.S * SLABF;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, a.S * SLABF * 4, 0x00020000);
        if (cons) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const f32x4 v = {acc[i][mt][4 * e4], acc[i][mt][4 * e4 + 1], acc[i][mt][4 * e4 + 2], acc[i][mt][4 * e4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                               (z * SLABF + ((((wave * NT + i) * MT + mt) * 4 + e4) * 256)) * 4 + lane * 16, 0, AUX_SC1);
                    }
            if (a.pro == 1 && tid < 64) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(stat + tid * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (z * SLABF + BM * BNp) * 4 + tid * 16, 0, AUX_SC1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's partials have been written through ...
        __syncthreads();                                        // ... every thread's have
        unsigned int *flag = reinterpret_cast<unsigned int *>(stat + 2 * BM);       // (16 floats between the statistics and `par`)
        if (tid == 0) *flag = atomicAdd(a.cnt + tile, 1u);
        __syncthreads();
        const bool last = (*flag == (unsigned int)(a.S - 1));
        if (wave == 0) WS_STAMP(62);                            // partials parked, arrival known
        if (!last) return;
        if (tid == 0) atomicExch(a.cnt + tile, 0u);             // ready for the next launch that uses this counter
        if (cons) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][mt][e] = 0.f;
            for (int zz = 0; zz < a.S; ++zz) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                rs, (zz * SLABF + ((((wave * NT + i) * MT + mt) * 4 + e4) * 256)) * 4 + lane * 16, 0, AUX_SC1));
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][mt][4 * e4 + e] += v[e];
                        }
            }
            if (a.pro == 1 && tid < 64) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                for (int zz = 0; zz < a.S; ++zz)
                    t += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (zz * SLABF + BM * BNp) * 4 + tid * 16, 0, AUX_SC1));
                *reinterpret_cast<f32x4 *>(stat + tid * 4) = t;
            }
        }
        __syncthreads();
    }

    if (a.pro == 1) {
        // (sum x, sum x^2) over K -> (rstd, -mean rstd): out = rstd acc - mean rstd colsum + bias
        if (tid < BM) {
            const float inv = 1.0f / (float)a.Ktot;
            const float mean = stat[2 * tid] * inv;
            const float var = fmaxf(stat[2 * tid + 1] * inv - mean * mean, 0.f);
            const float rstd = rsqrtf(var + a.eps);
            stat[2 * tid] = rstd;
            stat[2 * tid + 1] = -mean * rstd;
        }
        __syncthreads();
    }

    h16 *os = smem;
    if (y >= a.ytr) {
        // transposed part (V^T[sample][channel][token] for the flash kernel): staged channel-major -- a lane holds ONE token and
        // 16 channels, so consecutive lanes write consecutive tokens of a channel row -- and leaves as 16-byte pieces of 8 tokens
        // (T % 128 == 0: the tile lies in one sample, every row is valid)
        const int pt = BM + 8;
        if (cons) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int tp = (wave * NT + i) * 32;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int ch = tp + 8 * g4 + 4 * lh;
                    f32x4 bb = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) bb = *reinterpret_cast<const f32x4 *>(par + ch);
                    if (a.pro == 1) cs = *reinterpret_cast<const f32x4 *>(par + WS_PSTR + ch);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        float rsd = 1.f, nmr = 0.f;
                        if (a.pro == 1) { rsd = stat[(32 * mt + l32) * 2]; nmr = stat[(32 * mt + l32) * 2 + 1]; }
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            os[(tp + 8 * g4 + 4 * lh + e) * pt + 32 * mt + l32] = (h16)(acc[i][mt][4 * g4 + e] * rsd + nmr * cs[e] + bb[e]);
                    }
                }
            }
        }
        __syncthreads();
        if (cons) {
            const int b = m0 / a.T, tb = m0 - b * a.T;
            h16 *ob = a.outT + (long long)b * a.sT + (long long)(nb_p - a.tr_n0) * a.ldt + tb;
            constexpr int CPT = BM / 8;
            for (int idx = tid; idx < BNp * CPT; idx += nthr) {
                const int ch = idx / CPT, c = idx - ch * CPT;
                l2d_st8(ob + (long long)ch * a.ldt + c * 8, l2d_ld8(os + ch * pt + c * 8));
            }
        }
        return;
    }

    if (wave == 0) WS_STAMP(59);                               // (last block: slabs summed) epilogue starts
    const int pitch = BNo + 8;                                 // halfs; row stride = 16 B mod 32 B
    if (cons) {
        float rsd[MT], nmr[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            rsd[mt] = 1.f; nmr[mt] = 0.f;
            if (a.pro == 1) { rsd[mt] = stat[(32 * mt + l32) * 2]; nmr[mt] = stat[(32 * mt + l32) * 2 + 1]; }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int tp = (wave * NT + i) * 32;               // block-local packed column of the tile
            if (a.epi == 1) {
                // GEGLU: packed tile rows [0,8) value, [8,16) gate of channels c..c+7, [16,24) / [24,32) of c+8..c+15: register
                // groups (0,1) and (2,3) hold value / gate of the SAME channels in the same lane
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int ch = tp + 16 * g2 + 4 * lh;
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(par + ch), bg = *reinterpret_cast<const f32x4 *>(par + ch + 8);
                    f32x4 cv = {0.f, 0.f, 0.f, 0.f}, cg = {0.f, 0.f, 0.f, 0.f};
                    if (a.pro == 1) { cv = *reinterpret_cast<const f32x4 *>(par + WS_PSTR + ch); cg = *reinterpret_cast<const f32x4 *>(par + WS_PSTR + ch + 8); }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        h16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[i][mt][8 * g2 + e] * rsd[mt] + nmr[mt] * cv[e] + bv[e];
                            const float g = acc[i][mt][8 * g2 + 4 + e] * rsd[mt] + nmr[mt] * cg[e] + bg[e];
                            o[e] = (h16)(v * l2d_gelu(g));
                        }
                        *reinterpret_cast<h16x4 *>(os + (32 * mt + l32) * pitch + (tp >> 1) + 8 * g2 + 4 * lh) = o;
                    }
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int ch = tp + 8 * g4 + 4 * lh;
                    f32x4 bb = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) bb = *reinterpret_cast<const f32x4 *>(par + ch);
                    if (a.pro == 1) cs = *reinterpret_cast<const f32x4 *>(par + WS_PSTR + ch);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        f32x4 b2 = bb;
                        if (a.rowbias) b2 += *reinterpret_cast<const f32x4 *>(par + 2 * WS_PSTR + mt * WS_PSTR + ch);
                        h16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (h16)(acc[i][mt][4 * g4 + e] * rsd[mt] + nmr[mt] * cs[e] + b2[e]);
                        *reinterpret_cast<h16x4 *>(os + (32 * mt + l32) * pitch + tp + 8 * g4 + 4 * lh) = o;
                    }
                }
            }
        }
    }
    if (wave == 0) WS_STAMP(57);                               // tile staged (this wave)
    __syncthreads();
    if (wave == 0) WS_STAMP(58);                               // ... by every wave
    // whole rows, 16 bytes per lane; cc is the same for every row of a thread, which is what lets it keep per-channel GroupNorm sums
    // in registers.  One pass per sample that overlaps the tile (sample boundaries are multiples of 32 rows, RPP divides 32).
    const bool gn = a.gn1 != nullptr;
    const int rows = a.M - m0 < BM ? a.M - m0 : BM;
    const int segT = gn ? a.gnT : (1 << 30);
    float *red = reinterpret_cast<float *>(smem + BM * pitch);  // [consumer threads][8], behind the staged tile
    for (int smp = m0 / segT; (long long)smp * segT < m0 + rows; ++smp) {
        const int lo = smp * (long long)segT > m0 ? smp * segT - m0 : 0;
        const int hi = (long long)(smp + 1) * segT - m0 < rows ? (smp + 1) * segT - m0 : rows;
        float gs[4], gq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
        const h16x2 ones2 = {(h16)1.0f, (h16)1.0f};
        if (on) {
#pragma unroll
            for (int k = 0; k < EITMAX; ++k) {
                const int row = rr + k * RPP;
                if (row < lo || row >= hi) continue;
                h16x8 v = l2d_ld8(os + row * pitch + cc * 8);
                if (a.res) {
                    if (k < PF) v = v + resp[k < PF ? k : 0];
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
        }
        if (gn) {
            // statistics of what was just stored (the fp16 values the consumer GroupNorm will read), per channel pair, reduced to
            // the consumer's groups inside the block, two integer atomics per (consumer, overlapped group)
            if (cons) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = on ? gs[e] : 0.f; red[tid * 8 + 4 + e] = on ? gq[e] : 0.f; }
            }
            __syncthreads();
            float *chs1 = red + nthr * 8, *chs2 = chs1 + (BNo >> 1);
            if (tid < (BNo >> 1)) {
                const int c8 = tid >> 2, e = tid & 3;
                float s1 = 0.f, s2 = 0.f;
                for (int r = 0; r < RPP; ++r) { s1 += red[(r * CPR + c8) * 8 + e]; s2 += red[(r * CPR + c8) * 8 + 4 + e]; }
                chs1[tid] = s1; chs2[tid] = s2;
            }
            __syncthreads();
            l2d_gn_flush(a.gn1, a.gnG, a.cpg1 >> 1, a.choff1 >> 1, smp, chs1, chs2, nb_o >> 1, BNo >> 1, tid);
            l2d_gn_flush(a.gn2, a.gnG, a.cpg2 >> 1, a.choff2 >> 1, smp, chs1, chs2, nb_o >> 1, BNo >> 1, tid);
            __syncthreads();
        }
    }
    if (wave == 0) WS_STAMP(63);                               // rows stored
}

template <int NT, int RDS, int NL, bool NTW, int MAXW>
static void launch_ws(const WsArgs &a, int nthr, size_t lds, hipStream_t s) {
    static bool attr_done[L2D_MAX_DEV] = {false};
    const int dev = l2d_dev_ordinal();
    if (lds > 65536 && !attr_done[dev]) {   // > 64 KB of dynamic LDS: opted into once per kernel and device
        if (hipFuncSetAttribute((const void *)wsgemm_kernel<NT, RDS, NL, NTW, MAXW>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) == hipSuccess)
            attr_done[dev] = true;
        else
            (void)hipGetLastError();
    }
    hipLaunchKernelGGL((wsgemm_kernel<NT, RDS, NL, NTW, MAXW>), dim3(a.nm * a.ny * a.S), dim3(nthr), lds, s, a);
}

template <int NT, int RDS, int MAXW>
static void launch_ws_v(const WsArgs &a, int NL, bool ntw, int nthr, size_t lds, hipStream_t s) {
    if (NL == 1) {
        if (ntw) launch_ws<NT, RDS, 1, true, MAXW>(a, nthr, lds, s);
        else launch_ws<NT, RDS, 1, false, MAXW>(a, nthr, lds, s);
    } else {
        if (ntw) launch_ws<NT, RDS, 2, true, MAXW>(a, nthr, lds, s);
        else launch_ws<NT, RDS, 2, false, MAXW>(a, nthr, lds, s);
    }
}

int l2d_launch_wsgemm(const l2d_op *op, hipStream_t s) {
    WsArgs a;
    a.x1 = (const h16 *)op->p[0]; a.x2 = (const h16 *)op->p[1]; a.w = (const h16 *)op->p[2];
    a.bias = (const float *)op->p[3]; a.rowbias = (const float *)op->p[4]; a.res = (const h16 *)op->p[5];
    a.out = (h16 *)op->p[6]; a.zero = (const h16 *)op->p[7]; a.outT = (h16 *)op->p[8];
    a.gn1 = (unsigned long long *)op->p[9]; a.gn2 = (unsigned long long *)op->p[10];
    a.cnt = (unsigned int *)op->p[11]; a.ws = (float *)op->p[12]; a.colsum = (const float *)op->p[13];
    a.taps = op->i[0]; a.C1 = op->i[1]; a.C2 = op->i[2]; a.ldx1 = op->i[3]; a.ldx2 = op->i[4]; a.CinP = op->i[5];
    const int B = op->i[6];
    a.H = op->i[7]; a.W = op->i[8];
    const int NW = op->i[9], NT = op->i[10], NL = op->i[11] ? op->i[11] : 1;
    a.S = op->i[12] > 0 ? op->i[12] : 1;
    a.M = op->i[13];
    const int Nout = op->i[14];
    a.ldo = op->i[15]; a.ldr = op->i[16]; a.ldrb = op->i[17]; a.rows_per_bias = op->i[18]; a.epi = op->i[19];
    a.pro = op->i[20];
    const int ntr = op->i[21];                      // number of TRAILING 32-row weight tiles whose output is stored transposed
    a.ldt = op->i[22];
    const bool ntw = op->i[23] != 0;
    a.gnT = op->i[24]; a.gnG = op->i[25]; a.cpg1 = op->i[26]; a.choff1 = op->i[27]; a.cpg2 = op->i[28]; a.choff2 = op->i[29];
    a.T = op->i[30];
    a.sT = op->l[0];
    a.eps = op->f[0];
#ifdef L2D_PROBES
    a.probe = g_wsgemm_probe;
#endif
    if (!a.gn1 && a.gn2) { a.gn1 = a.gn2; a.cpg1 = a.cpg2; a.choff1 = a.choff2; a.gn2 = nullptr; }
    a.Ktot = a.taps * a.CinP;
    const int tiles = Nout > 0 ? Nout / 32 : 0;
    // NT = 2 (two weight tiles per consumer wave: half the activation re-reads from L2) runs with a register ring of 2 stages instead
    // of 4 and at most 4 consumer waves: 234 VGPRs, no scratch (its first form, ring of 4, ran AT the 256-register limit and was
    // removed in the middle of round 4; tests/test_kernel_resources.py replays this one like the others).
    const bool geom_ok = (NT == 1 || (NT == 2 && NW <= 4)) && NW >= 1 && NW <= 10 && (NL == 1 || NL == 2) && tiles > 0 &&
                         (Nout % 32) == 0 && (tiles % (NW * NT)) == 0 && ntr >= 0 && ntr <= tiles && (ntr % (NW * NT)) == 0;
    const bool conv = a.taps == 9;
    if (!a.x1 || !a.w || !a.zero || a.M <= 0 || a.M >= (1 << 22) || (a.taps != 1 && a.taps != 9) || !geom_ok || a.C1 <= 0 || (a.C1 % 64) ||
        a.C2 < 0 || (a.C2 % 64) || (a.C2 > 0 && !a.x2) || a.CinP != a.C1 + a.C2 || (a.ldx1 % 8) || a.ldx1 < a.C1 ||
        (a.C2 > 0 && ((a.ldx2 % 8) || a.ldx2 < a.C2)) || a.epi < 0 || a.epi > 1 || a.pro < 0 || a.pro > 1 ||
        (a.pro == 1 && (!a.colsum || conv)) || (a.epi == 1 && (!a.bias || a.res || ntr != 0 || a.rowbias)) ||
        (ntr < tiles && (!a.out || (a.ldo % 8))) || (a.res && (a.ldr % 8)) ||
        (conv && (B <= 0 || a.H <= 0 || a.W < 8 || a.H >= 32768 || a.W >= 32768 || a.M != B * a.H * a.W)) ||
        (a.rowbias && (a.ldrb <= 0 || a.rows_per_bias <= 0 || (a.rows_per_bias % 32))) ||
        (ntr > 0 && (!a.outT || a.T <= 0 || (a.T % 128) || (a.M % a.T) || (a.ldt % 8) || a.ldt < a.T)) ||
        (a.S > 1 && (!a.ws || !a.cnt || ntr != 0 || a.S > a.Ktot / 64)) || a.CinP > 32000 ||
        (((unsigned long long)a.x1 | (unsigned long long)a.x2 | (unsigned long long)a.w | (unsigned long long)a.out |
          (unsigned long long)a.res | (unsigned long long)a.outT | (unsigned long long)a.bias | (unsigned long long)a.colsum |
          (unsigned long long)a.rowbias | (unsigned long long)a.ws | (unsigned long long)a.zero) & 15)) {
        l2d_set_error("wsgemm(tag %d): invalid arguments (taps=%d M=%d C1=%d C2=%d CinP=%d Nout=%d ldo=%d epi=%d pro=%d NW=%d NT=%d NL=%d S=%d "
                      "ntr=%d T=%d H=%d W=%d)", op->tag, a.taps, a.M, a.C1, a.C2, a.CinP, Nout, a.ldo, a.epi, a.pro, NW, NT, NL, a.S, ntr,
                      a.T, a.H, a.W);
        return L2D_EINVAL;
    }
    if ((long long)(a.M + 128) * a.ldx1 * 2 >= (1ll << 31) || (a.C2 > 0 && (long long)(a.M + 128) * a.ldx2 * 2 >= (1ll << 31))) {
        l2d_set_error("wsgemm(tag %d): activation tensor too large for 32-bit row offsets (M=%d)", op->tag, a.M);
        return L2D_EINVAL;
    }
    const int BM = 128, BNp = NW * NT * 32, BNo = a.epi == 1 ? BNp / 2 : BNp, nthr = 64 * NW;
    if (a.gn1) {
        if (a.gnT <= 0 || (a.gnT % 32) || (a.M % a.gnT) || a.gnG <= 0 || a.gnG > 32 || a.cpg1 <= 0 || (a.gn2 && a.cpg2 <= 0) ||
            ((a.cpg1 | a.choff1) & 1) || (a.gn2 && ((a.cpg2 | a.choff2) & 1)) || ntr != 0 || a.epi == 1 || nthr < 32 * ((BNo / 2 + 31) / 32) ||
            nthr < BNo / 2) {
            l2d_set_error("wsgemm(tag %d): GroupNorm statistics need T %% 32 == 0 (T=%d), even group sizes and offsets, no transposed part",
                          op->tag, a.gnT);
            return L2D_EINVAL;
        }
    }
    if (!conv) { a.H = 1 << 14; a.W = 1 << 14; a.T = a.T > 0 ? a.T : a.M; }     // (linear: every "neighbour" is the token itself)
    else a.T = a.H * a.W;
    a.invT = 1.0f / (float)a.T; a.invW = 1.0f / (float)a.W;
    a.nm = (a.M + BM - 1) / BM;
    a.ny = tiles / (NW * NT);
    a.ytr = (tiles - ntr) / (NW * NT);
    a.tr_n0 = (tiles - ntr) * 32;
    const int nch = a.Ktot / 64;
    a.cps = nch / a.S; a.crem = nch % a.S;
    a.ncpt = a.CinP / 64;
    // LDS: 4-stage ring (64 KB) or, in the epilogue, the staged fp16 tile + GroupNorm reduction scratch; row statistics + flag behind both
    const size_t ring = (size_t)4 * BM * 64 * 2;
    size_t epi = (size_t)BM * (BNo + 8) * 2 + (a.gn1 ? (size_t)nthr * 32 + (size_t)BNo * 4 : 0);
    if (ntr > 0 && (size_t)BNp * (BM + 8) * 2 > epi) epi = (size_t)BNp * (BM + 8) * 2;
    const size_t body = ring > epi ? ring : epi;
    a.stat_off = (int)((body + 255) & ~(size_t)255);
    const size_t lds = (size_t)a.stat_off + 2 * BM * 4 + 64 + 6 * WS_PSTR * 4;      // + bias | column sums | 4 time-embedding rows
    if (lds > 163840 || (long long)a.nm * a.ny * a.S >= (1 << 24) || (long long)a.S * (BM * BNp + 2 * BM) * 4 >= (1ll << 31)) {
        l2d_set_error("wsgemm(tag %d): tile does not fit (LDS %zu bytes, %d x %d x %d blocks)", op->tag, lds, a.nm, a.ny, a.S);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    const int nthr_all = 64 * (NW + NL);
    if (NT == 2) launch_ws_v<2, 2, 6>(a, NL, ntw, nthr_all, lds, s);
    else if (NW <= 4) launch_ws_v<1, 4, 6>(a, NL, ntw, nthr_all, lds, s);
    else if (NW <= 8) launch_ws_v<1, 2, 10>(a, NL, ntw, nthr_all, lds, s);
    else launch_ws_v<1, 2, 12>(a, NL, ntw, nthr_all, lds, s);      // nine / ten consumer waves (3 / 3 / 2 / 2 consumers per SIMD beside two loaders)
    return l2d_check_launch("wsgemm", op->tag);
}
