// 3x3 stride-1 pad-1 convolution on MFMA for gfx950 with the activation PATCH resident in LDS and the WEIGHTS streamed straight
// into registers (round 6).  Replaces the InflatedConv3d 3x3 convs of ResnetBlock3D and Upsample3D (reference resnet.py:57-65,
// 94-127 (nearest x2 on (h, w), then the conv), 194, 214, 229-259) wherever a level has whole 8 x 16 pixel patches.
//
// Why a fourth conv form.  In-kernel stamps of the round-3 / round-4 kernels (profiles/round4_c_wsgemm_block_phases.txt, the
// per-stage cadence of pconv.hip) show every one of them paced by something other than the matrix cores:
//   * igemm.hip / wsgemm.hip gather a [tokens x 64] activation tile per (tap, channel chunk) stage: nine LDS-DMA passes over the
//     same pixels, and an LDS-DMA wave-instruction (1 KB) costs its issuing wave 90-180 cycles -- the loader waves pace the loop
//     (1 000 - 1 400 cycles per 512-MFMA-cycle stage);
//   * pconv.hip keeps the patch resident but DMAs the 64 x 64 weight tile of every stage through LDS (3 DMA instructions per wave
//     and stage) and feeds 16x16x32 MFMAs from 32-channel x 64-token wave tiles: 12 KB of LDS reads per 16 MFMAs;
//   * wsgemm.hip's consumers own 32 channels x 128 tokens: 4 KB of LDS reads per 4 MFMAs, one wave per SIMD, every read latency
//     exposed; its split-K tail is one block summing S x 64 KB slabs at ~65 GB/s.
// Here:
//   * a block owns one 8 x 16 patch (128 tokens) x BN = 64 CG output channels x one K slice (whole 64-channel chunks);
//   * the haloed patch (10 x 18 pixels x 64 channels = 23 KB) is DMA'd into LDS ONCE per chunk by dedicated loader wave(s), double
//     buffered, and serves all nine taps (a tap is an immediate offset of the fragment read); zero padding, the nearest-x2
//     up-sampling of Upsample3D and the channel concat of two inputs are resolved when the loader builds its per-lane pixel
//     offsets -- 24 DMA instructions per chunk instead of 144;
//   * weights never touch LDS: packed at load time in MFMA-fragment order per (64-channel tile, K group) as ONE sequential
//     stream (ops.pack_cconv), a compute wave pulls 2 KB per k step (two 32-row A fragments) through a 9-step register ring
//     (18 KB in flight per wave) with plain global loads the compiler counts exactly -- the chunk loop is rolled, its body (one
//     chunk = 9 taps) is straight-line code with static ring positions and immediate LDS offsets;
//   * a compute wave owns 64 channels x ALL 128 tokens (acc = 2 x 4 tiles of v_mfma_f32_32x32x16_f16): 4 KB of LDS reads + 2 KB of
//     weights per 8 MFMAs (256 matrix-pipe cycles) -- a quarter of wsgemm's LDS traffic per MFMA, a third of pconv's;
//   * K groups INSIDE the block: the KG waves that share a channel tile split every chunk by k step (group g takes the k steps
//     kk = g mod KG of each tap), all of them read the one resident patch; their partial accumulators meet through LDS once, at
//     the end (fixed order 0..KG-1), each wave keeping 4 / KG token tiles for the epilogue.  This fills four SIMDs from a
//     128 x 128 (CG = 2, KG = 2) or 128 x 64 (CG = 1, KG = 4) output tile without a larger tile or more split-K slabs;
//   * no block barrier inside a chunk (waves own disjoint accumulators and read-only LDS): one barrier per chunk = per 144 / KG ... MFMAs;
//   * split-K across blocks over whole chunks with the fused, fence-free reduction of igemm.hip / wsgemm.hip (write-through slabs,
//     arrival counter, last block sums in the fixed order 0..S-1: bit-repeatable);
//   * epilogue = pconv.hip's: fp32 bias + per-sample time-embedding row -> fp16 -> LDS -> whole rows (16 B per lane) with the
//     residual added in fp16, GroupNorm statistics of the output as fixed-point integer atomics.
// Patch image in LDS: pixel-major, 128 bytes (64 channels) per pixel, so one DMA instruction moves eight whole 128-byte lines (a
// slot-major image costs 64 lines of 16 useful bytes per instruction: measured ~150 cycles per instruction whatever the number of
// loaders); the 16-byte channel slot q of a pixel sits at position q ^ ((patch column >> 1) & 7).  The 32 tokens of an MFMA token tile
// are two patch rows of 16 pixels: with that key the 16 lanes of every ds_read_b128 lane group hit 16 distinct bank slots for every tap
// (checked exhaustively, tests/test_host_logic.py), and because the key depends on the column only, a tap's row offset and the token
// tile are IMMEDIATE offsets of the read: three base registers per k sub-step (one per dx), no address arithmetic in the loop.
// Rounding points: conv output (+ bias) -> fp16, residual add in fp16 (as igemm / pconv / wsgemm).
// Built WITHOUT packed fp32 VALU instructions like the rest of the library (csrc/Makefile).
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define L2D_GPTR(p) ((__attribute__((address_space(1))) const void *)(p))
#define L2D_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

struct CConvArgs {
    const h16 *x1, *x2, *w, *zero;
    const float *bias, *rowbias;
    const h16 *res;
    h16 *out;
    float *ws;                     // split-K slabs: [tile][S][128 * BN] floats
    unsigned int *cnt;             // split-K arrival counters, one per tile; zero before and after
    unsigned long long *gn1, *gn2;
    int B, H, W, Hs, Ws, ups;      // output resolution H x W; source resolution Hs x Ws (= H >> ups, W >> ups)
    int C1, C2, ldx1, ldx2, nch;   // nch: 64-channel chunks of the whole contraction
    int Nout, ldo, ldr, ldrb, rows_per_bias;
    int npx, npy, npat, ntn, S, cps, crem, nwg;
    int gnT, gnG, cpg1, choff1, cpg2, choff2;
    // GroupNorm (+ SiLU) of the INPUT fused into the loaders (pro = 1): statistics of the input from its producers' fixed-point atomics
    const long long *pacc;         // int64 [B][pG][2]: sum x in units of 2^-20, sum x^2 in units of 2^-12 (as L2D_OP_GN_APPLY with nchunk = 0)
    const h16 *pgamma, *pbeta;     // [C1 + C2]
    int pro, pG;
    float peps;
    int par_off;                   // byte offset of the epilogue parameters + flag word in LDS (launcher: behind patch buffers / tables / parked partials / staged tile)
#ifdef L2D_PROBES
    unsigned long long *probe;     // analysis builds: 32 s_memtime stamps per block (tools/cconv_stamps.py)
#endif
};

#ifdef L2D_PROBES
static unsigned long long *g_cconv_probe = nullptr;
extern "C" void l2d_cconv_set_probe(void *p) { g_cconv_probe = (unsigned long long *)p; }
#define CC_STAMP(i) do { if (a.probe && lane == 0) a.probe[(unsigned long long)blockIdx.x * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define CC_STAMP(i) do { } while (0)
#endif

namespace {
constexpr int CC_PH = 8, CC_PW = 16, CC_PWH = CC_PW + 2, CC_NPIX = (CC_PH + 2) * CC_PWH, CC_NPIXP = 192, CC_SEG = 3;
constexpr int CC_PBUFH = CC_NPIXP * 64;             // halfs per patch buffer: 192 pixels x 64 channels = 24 KB
constexpr int CC_RING = 9;                          // weight ring depth in k steps (2 KB each)
constexpr int CC_NBUF = 3;                          // patch buffers: the loaders run two chunks ahead of the compute waves
constexpr int CC_TBL_OFF = CC_NBUF * 24576;          // byte offset of the fused-GroupNorm (scale | shift) tables: one 512-byte entry per chunk of the slice
constexpr int CC_TBL_MAXCH = 48;                    // ... at most this many chunks per K slice when the input GroupNorm is fused
// (the epilogue parameters -- bias | time-embedding row -- and the flag word sit behind everything else: CConvArgs::par_off)
}  // namespace

// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, lane-linear at the LDS byte address in M0) as inline asm: with the builtin in
// the kernel hipcc's waitcnt pass tracks "LDS written by VMEM" and, at the head of the compute waves' rolled chunk loop, waits
// vmcnt(0) instead of the counted vmcnt(16) the weight ring needs (seen in the ISA: one drained ring per chunk).  The asm form is
// invisible to it; the loader waits for its own DMAs with an asm s_waitcnt.  M0 is saved and restored (the compiler owns it).
__device__ __forceinline__ void cc_dma16(unsigned long long gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void cc_dma4(unsigned long long gsrc, unsigned lds_dst) {       // 4 bytes per lane
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#ifndef CC_WNT
#define CC_WNT 0
#endif
__device__ __forceinline__ h16x8 cc_wload(const h16 *p) {
#if CC_WNT
    return __builtin_nontemporal_load(reinterpret_cast<const h16x8 *>(p));
#else
    return l2d_ld8(p);
#endif
}

// CG: 64-channel tiles per block; KG: K groups per channel tile; NLD: loader waves
template <int CG, int KG, int NLD>
__global__ __launch_bounds__(64 * (CG * KG + NLD), 2) void cconv_kernel(CConvArgs a) {     // (two waves per SIMD: <= 256 registers -- the four-wave forms share a CU two blocks at a time)
    constexpr int NCW = CG * KG, UPT = 4 / KG, SPC = 9 * UPT, R = CC_RING, OWN = 4 / KG, BN = 64 * CG, NTHR = 64 * NCW;
    static_assert(KG == 1 || KG == 2 || KG == 4, "K groups");
    static_assert(SPC % R == 0, "the ring position of a k step must be static");
    extern __shared__ __attribute__((aligned(16))) h16 smem[];     // patch[2] | (epilogue: K-group partials | staged tile | reduction)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool cons = wave < NCW;

    // XCD-aware bijective block order: an XCD runs a contiguous range of work items; items are patch-fastest, so the blocks that
    // share a weight band (same channel tile, same K slice) are neighbours on one XCD: the band leaves HBM once per XCD at most
    int wgid;
    {
        const int q = a.nwg >> 3, r = a.nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int rest = wgid / a.npat, pat = wgid - rest * a.npat;
    const int tile_n = rest / a.S, z = rest - tile_n * a.S;
    const int bb = pat / (a.npx * a.npy), pr = pat - bb * (a.npx * a.npy);
    const int py = pr / a.npx;
    const int y0 = py * CC_PH, x0 = (pr - py * a.npx) * CC_PW;
    const int c0 = z * a.cps + (z < a.crem ? z : a.crem);          // this block's K slice: chunks [c0, c0 + n)
    const int n = a.cps + (z < a.crem ? 1 : 0);
    const int n0 = tile_n * BN;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][mt][e] = 0.f;

    const int l32 = lane & 31, lh = lane >> 5;
    // this lane's token inside a 32-token tile: patch row (l32 >> 4) of the tile's two, pixel l32 & 15
    const int trow = l32 >> 4, tcol = l32 & 15;
    const int kgw = __builtin_amdgcn_readfirstlane(cons ? wave / CG : 0), cgw = __builtin_amdgcn_readfirstlane(cons ? wave - kgw * CG : 0);

    if (!cons) {
        // ---------------------------------------------------------------------------------------------- loader wave(s)
        // DMA instruction (q, seg) of a chunk moves pixels seg * 64 + lane of channel slot q (16 bytes per lane) to
        // patch[(q * 192 + pixel) * 16 B]; pixel p of the haloed patch = (row p / 18, column p % 18), image pixel (y0 - 1 + row,
        // x0 - 1 + column); pixels outside the image (the conv's zero padding) and lanes beyond the patch read the zero page.
        const int l = wave - NCW;
        __builtin_amdgcn_s_setprio(3);                       // few instructions, all of them on the compute waves' critical path: win every arbitration
        constexpr int DPC = 24 / NLD;                        // DMA instructions per loader and chunk
        static_assert(24 % NLD == 0, "every loader issues the same number of DMA instructions per chunk");
        // DMA instruction j (0..23) of a chunk moves the eight pixels 8 j .. 8 j + 7 of the haloed patch, one 128-byte line (64 channels)
        // each: lane -> (pixel 8 j + lane / 8, 16-byte position lane % 8); position pos of pixel p holds channel slot pos ^ key(p),
        // key = (patch column >> 1) & 7 -- the bank swizzle, applied to the SOURCE because the DMA image is lane-linear.  Pixel p =
        // (row p / 18, column p % 18) = image pixel (y0 - 1 + row, x0 - 1 + column); pixels outside the image (the conv's zero padding)
        // and beyond the patch read the zero page.  This loader owns instructions l DPC .. l DPC + DPC - 1.
        unsigned pix2[DPC], qb[DPC];                         // 2 x pixel index in the source image; byte offset of the channel slot
        bool okp[DPC];
#pragma unroll
        for (int jj = 0; jj < DPC; ++jj) {
            const int p = 8 * (l * DPC + jj) + (lane >> 3);
            const int r = p / CC_PWH, cx = p - r * CC_PWH;
            const int iy = y0 - 1 + r, ix = x0 - 1 + cx;
            okp[jj] = p < CC_NPIX && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
            pix2[jj] = okp[jj] ? 2u * (unsigned)((bb * a.Hs + (iy >> a.ups)) * a.Ws + (ix >> a.ups)) : 0u;
            qb[jj] = (unsigned)(((lane & 7) ^ ((cx >> 1) & 7)) * 16);
        }
        const int nc1 = a.C1 >> 6;
        const unsigned lds0 = (unsigned)(size_t)L2D_LPTR(smem);
        const unsigned long long zp = (unsigned long long)a.zero;
        auto issue_chunk = [&](int c, unsigned bufo) __attribute__((always_inline)) {      // chunk c0 + c -> patch buffer at byte offset bufo
            const int cc = c0 + c;
            const bool first = cc < nc1;
            const unsigned long long xb = first ? (unsigned long long)(a.x1 + cc * 64) : (unsigned long long)(a.x2 + (cc - nc1) * 64);   // wave-uniform
            const unsigned ldsel = (unsigned)(first ? a.ldx1 : a.ldx2);
            const unsigned dst = lds0 + bufo + (unsigned)l * (DPC * 1024);
#pragma unroll
            for (int jj = 0; jj < DPC; ++jj)                 // (32-bit byte offsets: checked by the launcher)
                cc_dma16(okp[jj] ? xb + (pix2[jj] * ldsel + qb[jj]) : zp, dst + jj * 1024);
        };
        if (l == 0) CC_STAMP(10);
        if (l == NLD - 1) {
            // epilogue parameters of this block's channels -> LDS (bias | this sample's time-embedding row): as cold global loads at the
            // START of the epilogue they would be a full round trip on the last arriver's critical path.  Older than every patch DMA of
            // this wave in the in-order queue, so the first counted wait covers them.
            const float *rb = a.rowbias ? a.rowbias + (long long)((bb * a.H * a.W) / a.rows_per_bias) * a.ldrb : nullptr;
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                if (a.bias) cc_dma4((unsigned long long)(a.bias + n0 + j * 64 + lane), lds0 + a.par_off + j * 256);
                if (rb) cc_dma4((unsigned long long)(rb + n0 + j * 64 + lane), lds0 + a.par_off + BN * 4 + j * 256);
            }
        }
        // The loaders run TWO chunks ahead (three patch buffers): the DMAs of chunk c + 2 are in flight while chunk c + 1 lands and chunk
        // c is consumed; a loader waits for "all but my DPC youngest" (in-order return), never for an empty queue inside the loop.
        issue_chunk(0, 0);
        if (n > 1) issue_chunk(1, CC_PBUFH * 2);
        // Fused input GroupNorm + SiLU (the reference's `conv(F.silu(norm(x)))`, resnet.py:233-234,249-250): the raw activations are DMA'd as
        // before; once a loader's share of a chunk has landed it normalises those very bytes in place -- y = silu(x scale[c] + shift[c]),
        // fp32, rounded to fp16 exactly where the separate GroupNorm launch rounded -- before the chunk is handed to the compute waves.
        // Pixels outside the image stay zero (the conv pads the NORMALISED tensor).  scale = rstd gamma, shift = beta - mean rstd gamma per
        // channel of this block's sample come from the producers' fixed-point sums; the tables of the slice's chunks are built once, here,
        // under the first DMAs (chunk c by loader c mod NLD, lane = channel).
        float *tbl = reinterpret_cast<float *>(reinterpret_cast<char *>(smem) + CC_TBL_OFF);
        if (a.pro) {
            const int cpg = (a.C1 + a.C2) / a.pG;
            const float inv = 1.0f / ((float)(a.Hs * a.Ws) * (float)cpg);
            for (int c = l; c < n; c += NLD) {
                const int ch = (c0 + c) * 64 + lane;
                const long long *src = a.pacc + ((long long)bb * a.pG + ch / cpg) * 2;
                const float sm = (float)((double)src[0] * (1.0 / 1048576.0));
                const float sq = (float)((double)src[1] * (1.0 / 4096.0));
                const float mean = sm * inv;
                const float var = fmaxf(sq * inv - mean * mean, 0.f);
                const float sc = rsqrtf(var + a.peps) * (float)a.pgamma[ch];
                tbl[c * 128 + lane] = sc;
                tbl[c * 128 + 64 + lane] = (float)a.pbeta[ch] - mean * sc;
            }
        }
        auto norm_chunk = [&](int c, unsigned bufo) __attribute__((always_inline)) {      // this loader's share of chunk c, in place
            h16 *base = smem + (bufo >> 1) + l * (DPC * 512) + lane * 8;
            const float *tc = tbl + c * 128;
            // all reads first (one LDS latency for the whole share, not one per instruction), branch-free arithmetic, a select keeps the
            // zeros of pixels outside the image / beyond the patch
            constexpr int NB = DPC > 6 ? 6 : DPC;             // instructions per batch (register budget)
#pragma unroll
            for (int j0 = 0; j0 < DPC; j0 += NB) {
                h16x8 v[NB];
                f32x4 s0[NB], s1[NB], h0[NB], h1[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int jj = j0 + k;
                    const int q8 = (int)(qb[jj] >> 1);        // first channel of this lane's 16-byte slot
                    v[k] = l2d_ld8(base + jj * 512);
                    s0[k] = *reinterpret_cast<const f32x4 *>(tc + q8); s1[k] = *reinterpret_cast<const f32x4 *>(tc + q8 + 4);
                    h0[k] = *reinterpret_cast<const f32x4 *>(tc + 64 + q8); h1[k] = *reinterpret_cast<const f32x4 *>(tc + 64 + q8 + 4);
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int jj = j0 + k;
                    h16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (h16)l2d_silu((float)v[k][e] * s0[k][e] + h0[k][e]);
                        o[4 + e] = (h16)l2d_silu((float)v[k][4 + e] * s1[k][e] + h1[k][e]);
                    }
                    l2d_st8(base + jj * 512, okp[jj] ? o : v[k]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        if (n > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.pro) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // (pro only, loaders AND compute waves: every loader's tables are written)
            asm volatile("" ::: "memory");
            norm_chunk(0, 0);
        }
        if (l == 0) CC_STAMP(12);
        __builtin_amdgcn_s_barrier();                        // chunk 0 has landed
        unsigned bufn = 2 * CC_PBUFH * 2, buf1 = CC_PBUFH * 2;   // buffers of chunk c + 2 / chunk c + 1
        for (int c = 0; c < n; ++c) {
            if (c + 2 < n) {
                issue_chunk(c + 2, bufn);                    // (its buffer was chunk c - 1's: every wave passed the previous barrier)
                bufn = bufn == (CC_NBUF - 1) * CC_PBUFH * 2 ? 0u : bufn + CC_PBUFH * 2;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPC) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (a.pro && c + 1 < n) norm_chunk(c + 1, buf1);
            buf1 = buf1 == (CC_NBUF - 1) * CC_PBUFH * 2 ? 0u : buf1 + CC_PBUFH * 2;
            if (l == 0 && c < 6) CC_STAMP(24 + c);           // chunk c + 1 landed (this loader's share)
            __builtin_amdgcn_s_barrier();                    // chunk c + 1 has landed; the compute waves are done with chunk c
        }
        if (l == 0) CC_STAMP(13);
    } else {
        // ---------------------------------------------------------------------------------------------- compute waves
        // weight stream of (channel tile n64, K group kg): [chunk][tap][u < UPT][half i < 2][64 lanes][8 halfs], k step kk = u KG + kg
        if (wave == 0) CC_STAMP(0);
        const int n64 = tile_n * CG + cgw;
        const h16 *wp = a.w + (((long long)n64 * KG + kgw) * a.nch + c0) * (SPC * 1024);
        const int wlane = lane * 8;
        h16x8 wr[R][2];
        long long wcur = 0;                                  // running wave-uniform element offset, opaque to the optimiser (rowgemm.hip)
#pragma unroll
        for (int s = 0; s < R; ++s) {
            asm volatile("" : "+s"(wcur));
#pragma unroll
            for (int i = 0; i < 2; ++i) wr[s][i] = cc_wload(wp + (wcur + i * 512) + wlane);
            wcur += 1024;
        }
        __builtin_amdgcn_sched_barrier(0);
        // this lane's fragment of k step kk = u KG + kg, tap (dy, dx), token tile mt: pixel p = (2 mt + trow + dy) 18 + tcol + dx, channel
        // slot q = 2 kk + lh at position q ^ key(column): lbase[dx][u] + (dy + 2 mt) * 18 * 64 halfs -- the key depends on dx only
        int lbase[3][UPT];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int u = 0; u < UPT; ++u)
                lbase[dx][u] = (trow * CC_PWH + tcol + dx) * 64 + (((2 * (u * KG + kgw) + lh) ^ (((tcol + dx) >> 1) & 7)) * 8);
        if (wave == 0) CC_STAMP(1);
        asm volatile("" ::: "memory");
        if (a.pro) __builtin_amdgcn_s_barrier();             // (the loaders' GroupNorm tables are complete: their barrier, met by every wave)
        __builtin_amdgcn_s_barrier();                        // chunk 0 has landed
        asm volatile("" ::: "memory");
        if (wave == 0) CC_STAMP(2);
        // One k step s of a chunk = (tap t = s / UPT, u = s % UPT): four fragment reads (one per token tile) + eight MFMAs.  The reads of
        // step s + 1 are issued in FRONT of step s's MFMAs (they fit under the previous step's last MFMA) -- across chunks too: the barrier
        // "chunk c + 1 has landed" is met at the START of chunk c's last step (every read of chunk c has returned by then: its buffer
        // is free for the loaders, who refill the one BEHIND it), and the next chunk's first fragments are fetched under that step's
        // MFMAs.  The refill of a weight-ring position follows the four MFMAs that consumed it.  One scheduling region per k step; the
        // body covers one chunk when a chunk is an even number of steps, two when odd (static fragment double-buffer parity).
        constexpr int CPB = (SPC & 1) ? 2 : 1;
        int bufo = 0;                                        // the current chunk's patch buffer (halfs)
        h16x8 xf[2][4];
        auto frag = [&](int s_, int mt) __attribute__((always_inline)) {       // (lbase[][] points into the CURRENT buffer)
            const int t = s_ / UPT, u = s_ - t * UPT;
            return l2d_ld8(smem + lbase[t % 3][u] + (t / 3 + 2 * mt) * (CC_PWH * 64));
        };
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xf[0][mt] = frag(0, mt);
#ifdef L2D_PROBES
        int cdone = 0;
#endif
        auto body = [&](auto ncc) __attribute__((always_inline)) {
            constexpr int NC = decltype(ncc)::value;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
#pragma unroll
                for (int s = 0; s < SPC; ++s) {
                    const int gs = j * SPC + s;
                    if (s + 1 == SPC) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of this chunk has returned
#ifdef L2D_PROBES
                        if (wave == 0 && cdone < 8) CC_STAMP(16 + cdone);    // chunk done (all but its last step's MFMAs)
                        ++cdone;
#endif
                        __builtin_amdgcn_s_barrier();            // this chunk's buffer is free; the next chunk has landed
                        asm volatile("" ::: "memory");
                        const int delta = bufo == (CC_NBUF - 1) * CC_PBUFH ? -(CC_NBUF - 1) * CC_PBUFH : CC_PBUFH;      // wave-uniform
                        bufo += delta;
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                            for (int u = 0; u < UPT; ++u) lbase[dx][u] += delta;
                    }
                    // (behind the slice's last chunk these read a buffer nobody filled: never used)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) xf[(gs + 1) & 1][mt] = frag(s + 1 == SPC ? 0 : s + 1, mt);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[i][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[gs % R][i], xf[gs & 1][mt], acc[i][mt], 0, 0, 0);
                        if (i == 0) asm volatile("" : "+s"(wcur));
                        wr[gs % R][i] = cc_wload(wp + (wcur + i * 512) + wlane);   // (beyond the slice: the next slice's / the pad's bytes, never used)
                    }
                    wcur += 1024;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        int c = 0;
        for (; c + CPB <= n; c += CPB) body(std::integral_constant<int, CPB>{});
        if constexpr (CPB == 2) {
            if (c < n) body(std::integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the trailing prefetch)
        if (wave == 0) CC_STAMP(3);
    }

    // ------------------------------------------------------------------------------------------------- K groups meet
    // Every compute wave keeps OWN = 4 / KG token tiles (mt = kg + KG m) and parks the others in LDS for their owners; an owner
    // sums the KG partials of a tile in the fixed order 0..KG-1 (its own at its position).  The patch buffers are dead.
    f32x16 own[2][OWN];
    float *red = reinterpret_cast<float *>(smem);
    constexpr int NPARK = 4 - OWN;                           // parked tiles per wave
    // parked tile mt of the wave of K group k is that wave's j-th parked tile: j = mt - #(tiles below mt that k owns)
    auto park_idx = [](int mt, int k) { int own_below = 0; for (int m = k; m < mt; m += KG) ++own_below; return mt - own_below; };
    if constexpr (KG > 1) {
        if (cons) {
            auto park = [&](auto kc) {
                constexpr int K_ = decltype(kc)::value;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (mt % KG == K_) continue;
                    const int blk = ((cgw * KG + K_) * NPARK + park_idx(mt, K_)) * 2;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const f32x4 v = {acc[i][mt][4 * e4], acc[i][mt][4 * e4 + 1], acc[i][mt][4 * e4 + 2], acc[i][mt][4 * e4 + 3]};
                            *reinterpret_cast<f32x4 *>(red + ((blk + i) * 4 + e4) * 256 + lane * 4) = v;
                        }
                }
            };
            if (kgw == 0) park(std::integral_constant<int, 0>{});
            else if (kgw == 1) park(std::integral_constant<int, 1>{});
            else if (KG > 2 && kgw == 2) park(std::integral_constant<int, 2 % KG>{});
            else if (KG > 2) park(std::integral_constant<int, 3 % KG>{});
        }
        __syncthreads();
        if (cons) {
            auto gather = [&](auto kc) {
                constexpr int K_ = decltype(kc)::value;
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    const int mt = K_ + KG * m;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        f32x16 tot;
#pragma unroll
                        for (int k = 0; k < KG; ++k) {
                            f32x16 p;
                            if (k == K_) p = acc[i][mt];
                            else {
                                const int blk = ((cgw * KG + k) * NPARK + park_idx(mt, k)) * 2;
#pragma unroll
                                for (int e4 = 0; e4 < 4; ++e4) {
                                    const f32x4 v = *reinterpret_cast<const f32x4 *>(red + ((blk + i) * 4 + e4) * 256 + lane * 4);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) p[4 * e4 + e] = v[e];
                                }
                            }
                            if (k == 0) tot = p;
                            else tot += p;
                        }
                        own[i][m] = tot;
                    }
                }
            };
            if (kgw == 0) gather(std::integral_constant<int, 0>{});
            else if (kgw == 1) gather(std::integral_constant<int, 1>{});
            else if (KG > 2 && kgw == 2) gather(std::integral_constant<int, 2 % KG>{});
            else if (KG > 2) gather(std::integral_constant<int, 3 % KG>{});
        }
        __syncthreads();                                     // (the parked partials are dead: the staged tile may overwrite them)
        if (wave == 0) CC_STAMP(4);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int m = 0; m < OWN; ++m) own[i][m] = acc[i][m];
    }

    const int tile = tile_n * a.npat + pat;
    float *par = reinterpret_cast<float *>(reinterpret_cast<char *>(smem) + a.par_off);      // bias [BN] | time-embedding row [BN]
    unsigned int *flag = reinterpret_cast<unsigned int *>(par + 2 * BN);
    if (a.S > 1) {
        // Split-K, reduction fused, no fences (igemm.hip explains why), and the reducing block neither publishes nor re-reads its own
        // partial tile: a block first draws a TICKET; the S - 1 blocks that do not draw the last one park their partial tiles in the
        // tile's slab with write-through (sc1) stores, drain them, and count themselves DONE; the block with the last ticket -- every
        // other slice of its tile is by then inside its epilogue, i.e. resident and past its last dependence on anything, so waiting
        // for them cannot deadlock whatever else shares the GPU -- polls DONE (one lane, relaxed, sleeping), then sums the slabs in the
        // fixed order 0..S-1 with its own registers at its own position: bit-repeatable whoever reduces.  It leaves both counters zero.
        constexpr int AUX_SC1 = 16;
        constexpr int SLABF = 128 * BN;
        float *slab = a.ws + (long long)tile * a.S * SLABF;
        unsigned int *tk = a.cnt + 2 * tile;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, a.S * SLABF * 4, 0x00020000);
        // every block starts parking its partial tile at once -- the stores are asynchronous and the ticket's round trip runs beside them;
        // the block that turns out to hold the last ticket simply does not wait for them (its slab is never read)
        if (tid == 0) *flag = atomicAdd(tk, 1u);
        if (cons) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < OWN; ++m)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const f32x4 v = {own[i][m][4 * e4], own[i][m][4 * e4 + 1], own[i][m][4 * e4 + 2], own[i][m][4 * e4 + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                               (z * SLABF + ((((wave * 2 + i) * OWN + m) * 4 + e4) * 256)) * 4 + lane * 16, 0, AUX_SC1);
                    }
        }
        __syncthreads();
        const bool last = (*flag == (unsigned int)(a.S - 1));
        if (!last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's partials have been written through ...
            __syncthreads();                                    // ... every thread's have
            if (tid == 0) atomicAdd(tk + 1, 1u);
            if (wave == 0) CC_STAMP(5);
            return;
        }
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(tk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned int)(a.S - 1) && ++spins < (1u << 24))
                __builtin_amdgcn_s_sleep(4);
            atomicExch(tk, 0u);                                 // ready for the next launch that uses these counters
            atomicExch(tk + 1, 0u);
        }
        __syncthreads();
        if (wave == 0) CC_STAMP(5);
        if (cons) {
            // the other slices' slabs, ZB at a time (one round trip per batch, not per slab), summed in slice order with this block's
            // registers at position z
            constexpr int ZB = KG == 4 ? 4 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x16 tot[OWN];
#pragma unroll
                for (int m = 0; m < OWN; ++m)
#pragma unroll
                    for (int e = 0; e < 16; ++e) tot[m][e] = 0.f;
                for (int z0 = 0; z0 < a.S; z0 += ZB) {
                    f32x4 v[ZB][OWN][4];
#pragma unroll
                    for (int b = 0; b < ZB; ++b) {
                        const int zz = z0 + b;
                        const int zr = (zz < a.S && zz != z) ? zz : (z == 0 ? (a.S > 1 ? 1 : 0) : 0);      // (clamped: an unconditional load of a valid slab)
#pragma unroll
                        for (int m = 0; m < OWN; ++m)
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4)
                                v[b][m][e4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rs, (zr * SLABF + ((((wave * 2 + i) * OWN + m) * 4 + e4) * 256)) * 4 + lane * 16, 0, AUX_SC1));
                    }
#pragma unroll
                    for (int b = 0; b < ZB; ++b) {
                        const int zz = z0 + b;
                        if (zz >= a.S) break;
                        if (zz == z) {
#pragma unroll
                            for (int m = 0; m < OWN; ++m) tot[m] += own[i][m];
                        } else {
#pragma unroll
                            for (int m = 0; m < OWN; ++m)
#pragma unroll
                                for (int e4 = 0; e4 < 4; ++e4)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) tot[m][4 * e4 + e] += v[b][m][e4][e];
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < OWN; ++m) own[i][m] = tot[m];
            }
        }
        if (wave == 0) CC_STAMP(6);
    }

    // ------------------------------------------------------------------------------------------------- epilogue
    // residual rows first (one latency under the staging): thread -> (row cidx / CPRN, 16-byte chunk cidx % CPRN)
    constexpr int CPRN = BN / 8, EPI_IT = (128 * CPRN) / NTHR;
    static_assert(NTHR % CPRN == 0, "a thread keeps one channel chunk");
    auto mrow = [&](int row) { return ((long long)bb * a.H + y0 + (row >> 4)) * a.W + x0 + (row & 15); };
    h16x8 resv[EPI_IT];
    if (cons && a.res) {
#pragma unroll
        for (int it = 0; it < EPI_IT; ++it) {
            const int cidx = it * NTHR + tid, row = cidx / CPRN, cc = cidx - row * CPRN;
            resv[it] = l2d_ld8(a.res + mrow(row) * a.ldr + n0 + cc * 8);
        }
    }
    constexpr int pitch = BN + 8;
    h16 *ot = smem;
    if (cons) {
        // D layout of the 32x32 MFMA: lane -> token (l32 of the tile), register r -> channel (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ch = cgw * 64 + i * 32 + 8 * g4 + 4 * lh;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (a.bias) bv = *reinterpret_cast<const f32x4 *>(par + ch);
                if (a.rowbias) bv += *reinterpret_cast<const f32x4 *>(par + BN + ch);
#pragma unroll
                for (int m = 0; m < OWN; ++m) {
                    const int mt = kgw + KG * m;
                    const int row = (mt * 2 + trow) * 16 + tcol;
                    h16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (h16)(own[i][m][4 * g4 + e] + bv[e]);
                    *reinterpret_cast<h16x4 *>(ot + row * pitch + ch) = o;
                }
            }
        }
    }
    __syncthreads();
    if (wave == 0) CC_STAMP(7);
    const bool gn = a.gn1 != nullptr;
    float gs[4], gq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
    const h16x2 ones2 = {(h16)1.0f, (h16)1.0f};
    if (cons) {
#pragma unroll
        for (int it = 0; it < EPI_IT; ++it) {
            const int cidx = it * NTHR + tid, row = cidx / CPRN, cc = cidx - row * CPRN;
            h16x8 v = l2d_ld8(ot + row * pitch + cc * 8);
            if (a.res) v = v + resv[it];
            l2d_st8(a.out + mrow(row) * a.ldo + n0 + cc * 8, v);
            if (gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const h16x2 pr2 = {v[2 * e], v[2 * e + 1]};
                    gs[e] = __builtin_amdgcn_fdot2(pr2, ones2, gs[e], false);
                    gq[e] = __builtin_amdgcn_fdot2(pr2, pr2, gq[e], false);
                }
            }
        }
    }
    if (gn) {
        __syncthreads();                                            // every thread is done reading the staged tile
        float *rd = reinterpret_cast<float *>(smem);                // [NTHR][8]: 4 pair sums | 4 pair sums of squares
        if (cons) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { rd[tid * 8 + e] = gs[e]; rd[tid * 8 + 4 + e] = gq[e]; }
        }
        __syncthreads();
        float *chs1 = rd + NTHR * 8, *chs2 = chs1 + BN / 2;
        if (tid < BN / 2) {
            const int cc = tid >> 2, e = tid & 3;
            float s = 0.f, q = 0.f;
            for (int r = 0; r < NTHR / CPRN; ++r) { s += rd[(r * CPRN + cc) * 8 + e]; q += rd[(r * CPRN + cc) * 8 + 4 + e]; }
            chs1[tid] = s; chs2[tid] = q;
        }
        __syncthreads();
        l2d_gn_flush(a.gn1, a.gnG, a.cpg1 >> 1, a.choff1 >> 1, bb, chs1, chs2, n0 >> 1, BN >> 1, tid);
        l2d_gn_flush(a.gn2, a.gnG, a.cpg2 >> 1, a.choff2 >> 1, bb, chs1, chs2, n0 >> 1, BN >> 1, tid);
    }
    if (wave == 0) CC_STAMP(8);
}

template <int CG, int KG, int NLD>
static void launch_cc(const CConvArgs &a, size_t lds, hipStream_t s) {
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (lds > 65536 && !attr_done) {
        if (hipFuncSetAttribute((const void *)cconv_kernel<CG, KG, NLD>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) == hipSuccess) attr_done = true;
        else (void)hipGetLastError();
    }
    hipLaunchKernelGGL((cconv_kernel<CG, KG, NLD>), dim3(a.nwg), dim3(64 * (CG * KG + NLD)), lds, s, a);
}

int l2d_launch_cconv(const l2d_op *op, hipStream_t s) {
    CConvArgs a;
    a.x1 = (const h16 *)op->p[0]; a.x2 = (const h16 *)op->p[1]; a.w = (const h16 *)op->p[2];
    a.bias = (const float *)op->p[3]; a.rowbias = (const float *)op->p[4]; a.res = (const h16 *)op->p[5];
    a.out = (h16 *)op->p[6]; a.zero = (const h16 *)op->p[7];
    a.gn1 = (unsigned long long *)op->p[9]; a.gn2 = (unsigned long long *)op->p[10];
    a.cnt = (unsigned int *)op->p[11]; a.ws = (float *)op->p[12];
    a.C1 = op->i[1]; a.C2 = op->i[2]; a.ldx1 = op->i[3]; a.ldx2 = op->i[4];
    const int CinP = op->i[5];
    a.B = op->i[6]; a.H = op->i[7]; a.W = op->i[8];
    const int CG = op->i[9], KG = op->i[10], NLD = op->i[11] ? op->i[11] : 1;
    a.S = op->i[12] > 0 ? op->i[12] : 1;
    a.ups = op->i[13] ? 1 : 0;
    a.Nout = op->i[14]; a.ldo = op->i[15]; a.ldr = op->i[16]; a.ldrb = op->i[17]; a.rows_per_bias = op->i[18];
    a.gnT = op->i[24]; a.gnG = op->i[25]; a.cpg1 = op->i[26]; a.choff1 = op->i[27]; a.cpg2 = op->i[28]; a.choff2 = op->i[29];
    a.pacc = (const long long *)op->p[13]; a.pgamma = (const h16 *)op->p[14]; a.pbeta = (const h16 *)op->p[15];
    a.pro = op->i[20]; a.pG = op->i[21]; a.peps = op->f[0];
#ifdef L2D_PROBES
    a.probe = g_cconv_probe;
#endif
    if (!a.gn1 && a.gn2) { a.gn1 = a.gn2; a.cpg1 = a.cpg2; a.choff1 = a.choff2; a.gn2 = nullptr; }
    const bool geo = ((CG == 2 && KG == 2) || (CG == 1 && KG == 4)) && (NLD == 1 || NLD == 2 || NLD == 4);
    if (!a.x1 || !a.w || !a.out || !a.zero || !geo || a.B <= 0 || a.H <= 0 || a.W <= 0 || (a.H % CC_PH) || (a.W % CC_PW) ||
        (a.ups && ((a.H | a.W) & 1)) || a.C1 <= 0 || (a.C1 % 64) || a.C2 < 0 || (a.C2 % 64) || (a.C2 > 0 && !a.x2) || CinP != a.C1 + a.C2 ||
        a.Nout <= 0 || (a.Nout % (64 * CG)) || (a.ldx1 % 8) || a.ldx1 < a.C1 || (a.C2 > 0 && ((a.ldx2 % 8) || a.ldx2 < a.C2)) ||
        (a.ldo % 8) || a.ldo < a.Nout || (a.res && ((a.ldr % 8) || a.ldr < a.Nout)) || (a.rowbias && (a.ldrb <= 0 || a.rows_per_bias <= 0)) ||
        a.S > CinP / 64 || (a.S > 1 && (!a.ws || !a.cnt)) || a.pro < 0 || a.pro > 1 ||
        (a.pro && (!a.pacc || !a.pgamma || !a.pbeta || a.ups || a.pG <= 0 || (CinP % a.pG) || !(a.peps > 0.f) ||
                   (CinP / 64 + a.S - 1) / a.S > CC_TBL_MAXCH || (((unsigned long long)a.pacc | (unsigned long long)a.pgamma | (unsigned long long)a.pbeta) & 7))) ||
        (((unsigned long long)a.x1 | (unsigned long long)a.x2 | (unsigned long long)a.w | (unsigned long long)a.out |
          (unsigned long long)a.res | (unsigned long long)a.bias | (unsigned long long)a.rowbias | (unsigned long long)a.ws |
          (unsigned long long)a.zero) & 15)) {
        l2d_set_error("cconv(tag %d): invalid arguments (B=%d H=%d W=%d ups=%d C1=%d C2=%d CinP=%d Nout=%d CG=%d KG=%d NLD=%d S=%d)", op->tag,
                      a.B, a.H, a.W, a.ups, a.C1, a.C2, CinP, a.Nout, CG, KG, NLD, a.S);
        return L2D_EINVAL;
    }
    if (a.gn1) {
        if (a.gnT != a.H * a.W || a.gnG <= 0 || a.gnG > 32 || a.cpg1 <= 0 || (a.gn2 && a.cpg2 <= 0) || ((a.cpg1 | a.choff1) & 1) ||
            (a.gn2 && ((a.cpg2 | a.choff2) & 1))) {
            l2d_set_error("cconv(tag %d): GroupNorm statistics need T == H * W, even group sizes and offsets", op->tag);
            return L2D_EINVAL;
        }
    }
    a.Hs = a.H >> a.ups; a.Ws = a.W >> a.ups;
    a.nch = CinP / 64;
    a.npx = a.W / CC_PW; a.npy = a.H / CC_PH; a.npat = a.B * a.npx * a.npy; a.ntn = a.Nout / (64 * CG);
    a.cps = a.nch / a.S; a.crem = a.nch % a.S;
    const long long nwg = (long long)a.npat * a.ntn * a.S;
    // (pixel indices are 32-bit, every element offset is formed in 64 bits; the slab of a tile is addressed with 32-bit byte offsets)
    if ((long long)a.B * a.H * a.W >= (1ll << 28) || nwg >= (1ll << 24) || (long long)a.S * 128 * 64 * CG * 4 >= (1ll << 31)) {
        l2d_set_error("cconv(tag %d): tensor too large for the kernel's index arithmetic", op->tag);
        return L2D_EINVAL;
    }
    a.nwg = (int)nwg;
    // LDS: three patch buffers (+ the GroupNorm tables of a fused prologue), reused by the K groups' parked partials, the staged tile and
    // the GroupNorm reduction; parameters + flag behind the largest of them.
    {
        const int NCW = CG * KG, BN = 64 * CG, OWNT = 4 / KG;
        size_t body = (size_t)CC_NBUF * 24576 + (a.pro ? (size_t)((a.nch + a.S - 1) / a.S) * 512 : 0);
        const size_t park = KG > 1 ? (size_t)NCW * (4 - OWNT) * 2 * 4096 : 0;
        const size_t stage = (size_t)128 * (BN + 8) * 2;
        const size_t gnred = (size_t)(NCW * 64 * 8 + BN) * 4;
        if (park > body) body = park;
        if (stage > body) body = stage;
        if (gnred > body) body = gnred;
        a.par_off = (int)((body + 255) & ~(size_t)255);
    }
    const size_t lds = (size_t)a.par_off + 2 * 64 * CG * 4 + 64;
    L2D_DRY_RETURN();
#define CC_LAUNCH(cg, kg) do { if (NLD == 1) launch_cc<cg, kg, 1>(a, lds, s); else if (NLD == 2) launch_cc<cg, kg, 2>(a, lds, s); \
                               else launch_cc<cg, kg, 4>(a, lds, s); } while (0)
    // ((CG 1, KG 2) -- two compute waves per block, two blocks per CU -- was built and measured in round 6: 32.0 us on the level-0
    //  conv against 27.9 for (1, 4) and 30.4 for pconv: not instantiated)
    if (CG == 2) CC_LAUNCH(2, 2);
    else CC_LAUNCH(1, 4);
#undef CC_LAUNCH
    return l2d_check_launch("cconv", op->tag);
}
