// Implicit-GEMM on MFMA for gfx950: nn.Linear / 1x1 conv / 3x3 conv (stride 1|2, nearest-2x upsample and
// channel-concat folded into the gather) with fused epilogue (bias, per-sample time-embedding bias,
// residual, GEGLU, SiLU).  Replaces the cuDNN/cuBLAS calls behind the reference's InflatedConv3d / nn.Linear
// (reference resnet.py:57-65,112,141,194,214,227; attention.py:62,89; motion_module.py:182,207).
//
//   out[m][n] = epi( sum_k X[m][k] * W[n][k] ),   m = (b, oy, ox) token, n = output channel,
//   k = (tap, ci) with ci contiguous (channels-last activations, weights packed [n][tap][CinP]).
//
// MI355X mapping
//   * v_mfma_f32_16x16x32_f16, fp32 accumulate.  The WEIGHT tile is the MFMA A operand (rows) and the
//     TOKEN tile the B operand (cols): the C/D layout (col = lane&15, row = 4*(lane>>4)+reg) then gives
//     every lane 4 consecutive output channels of one token -> 8-byte stores and vector bias/residual
//     loads in the epilogue; for GEGLU the value and gate columns (interleaved by 16 at pack time) land
//     in the same lane.
//   * 256 threads = 4 waves (2x2), block tile TN x TM (128x128 or 64x64).
//   * Operand staging is LDS-DMA: `global_load_lds_dwordx4` (16 B/lane, no VGPR round trip) into a ring of NS stages
//     of BK K-elements, counted `s_waitcnt vmcnt((NS-2)*LPS)` + raw `s_barrier`: NS-1 stages stay in flight and the
//     DMA queue is never drained inside the loop (one barrier per K step).  Pipeline shapes (op.i[23], launch_p):
//     BK32 x 3/4/6, BK64 x 2/3/4/6, BK128 x 2/3.  Warm, isolated sweeps are insensitive to the depth; in the frame the
//     weights arrive cold and the 64x64 tile wants BK64 x 3 (85.3 vs 79.9 frames/s with x2).  The per-shape choice of
//     (tile, split-K, shape) is a table measured in the frame (live2diff_amd/igemm_tuned.json, tools/igemm_pick.py).
//   * The LDS image of a DMA is lane-linear (wave-uniform base + 16*lane), so the bank swizzle is applied on
//     the SOURCE side: a lane fetches logical 16-byte slot  pslot ^ swz(row)  of its row and the fragment
//     reads apply the same XOR -> conflict-free ds_read_b128 (SQ_LDS_BANK_CONFLICT = 0 in the PMC pass).
//     Conv zero padding / ragged rows / channel tails read from a 16-byte zero page.
//   * PMC (profiles/r1b_pmc_ops.txt) showed ~12 VALU + 7 SALU instructions per MFMA in the first version, i.e. the loop
//     was bound by the gather's address arithmetic, not by the matrix cores.  The gather is therefore reduced to one
//     64-bit add + select per row per stage: per-row element offsets and a 9-bit tap-validity mask are computed once,
//     the per-stage part (tap offset, channel offset) is wave-uniform scalar work.  Nearest-upsample convs and
//     3x3 convs over a two-input concat use the generic (slower) gather.
//   * Split-K (grid.y) for the low-resolution levels (M = 128..2048, K up to 23 040): fp32 partial tiles to a
//     workspace; the block that arrives last at the tile's counter reduces them in a fixed order and runs the fused
//     epilogue (round 2; the separate `igemm_splitk_epilogue` launch of round 1 remains as the A/B path, cnt = 0).
//   * XCD-aware block order: each XCD runs a contiguous tile range, token-tile major or (op.i[22] & 16) weight-tile
//     major, whichever keeps the larger operand in ONE L2 (28.5 vs 43.6 MB of L2 fills per launch, frame average).
//   * In-kernel timestamps (s_memtime) on the GEGLU GEMM M 8192 x N 2560 x K 320: a 128x128 block lives 25.8 k cycles,
//     4.5 k before the loop (kernel arguments, descriptors, first DMA issue), 12.2 k in 10 K steps with three blocks
//     sharing the CU, 9.0 k in the epilogue -- the K = 320..1280 linear layers of this UNet are bound by what surrounds
//     the MFMA loop.
#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define L2D_GPTR(p) ((__attribute__((address_space(1))) const void *)(p))
#define L2D_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

struct IGemmArgs {
    // Field order = order of first use in the kernel: the argument block is fetched through the scalar cache in 64-byte lines,
    // every line touched for the first time is one more dependent miss (~500 cycles) on the block's critical path.
    // -- line 0: what the first weight DMA needs
    const h16 *w;
    const h16 *zero;   // >= 16 bytes of zeros
    long long sw;
    int Nout, Kp, nwg, ntn, ntm, order, splitk;      // nwg = ntn * ntm = gridDim.x (tiles), splitk = gridDim.y
    float inv_ntn, inv_ntm, inv_s;                    // reciprocals for the exact float-assisted divisions (l2d_divf)
    // -- token-row descriptors
    const h16 *x1, *x2;
    long long sx1;
    int M, C1, C2, ldx1, ldx2, CinP, B, Hin, Win, Hout, Wout, stride, ups, taps;
    int pad;   // low-side zero padding of the 3x3 gather: 1 (symmetric, nn.Conv2d padding=1) or 0 (TF-"SAME" of a stride-2 conv on an even size)
    float inv_hw, inv_wout;
    // -- epilogue
    const float *bias, *rowbias;
    const h16 *res;
    h16 *out;
    long long so, sres;
    int ldo, ldr, ldrb, rows_per_bias, epi, epl;
    float *ws;         // split-K workspace fp32: [S][M][NoutP] (two-launch reduction) or [tile][S][TN*TM] (fused, cnt != 0)
    unsigned int *cnt; // fused split-K reduction: one arrival counter per (batch, tile), zero before and after every launch
    // GroupNorm statistics of the output for up to two consumer GroupNorms (0 = none): fixed-point int64 accumulators
    // [sample][G][2]; T tokens per sample; (channels per group, channel offset in the consumer's concatenated axis) each
    unsigned long long *gn1, *gn2;
    int gnT, gnG, cpg1, choff1, cpg2, choff2;
#ifdef L2D_PROBES
    unsigned long long *probe;   // analysis builds: 8 s_memtime stamps per block (thread 0), see tools/igemm_probe.py
#endif
};

#ifdef L2D_PROBES
static unsigned long long *g_igemm_probe = nullptr;
extern "C" void l2d_igemm_set_probe(void *p) { g_igemm_probe = (unsigned long long *)p; }
#define L2D_STAMP(i)                                                                                              \
    do {                                                                                                          \
        if (a.probe && threadIdx.x == 0)                                                                          \
            a.probe[((blockIdx.z * gridDim.y + blockIdx.y) * (unsigned long long)gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define L2D_STAMP(i) do { } while (0)
#endif

// n / d for 0 <= n < 2^24, d >= 1, inv = 1.0f / d: float estimate (within one of the quotient) + one correction step either
// way -- ~8 instructions instead of the ~35 of the compiler's 32-bit division sequence (or ~150 for a 64-bit one)
__device__ __forceinline__ int l2d_divf(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    const int r = n - q * d;
    q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    return q;
}

// fused epilogue for 4 consecutive output channels n..n+3 of token m
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs &a, h16 *outp, const h16 *resp, int m, int n, f32x4 v) {
    const float *rb = a.rowbias ? a.rowbias + (long long)(m / a.rows_per_bias) * a.ldrb : nullptr;
    if (n + 4 > a.Nout) {
        // ragged last channel group (Nout % 4 != 0, e.g. the swapped V^T GEMM with an odd token count)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= a.Nout) break;
            float y = v[r];
            if (a.bias) y += a.bias[n + r];
            if (rb) y += rb[n + r];
            if (a.epi == 2) y = l2d_silu(y);
            if (a.epi == 3) y = fmaxf(y, 0.f);
            if (a.epi == 5) y = l2d_gelu(y);
            if (resp) y += (float)resp[(long long)m * a.ldr + n + r];
            if (a.epi == 4) y = fmaxf(y, 0.f);
            outp[(long long)m * a.ldo + n + r] = (h16)y;
        }
        return;
    }
    if (a.bias) v += *reinterpret_cast<const f32x4 *>(a.bias + n);
    if (rb) v += *reinterpret_cast<const f32x4 *>(rb + n);
    if (a.epi == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = l2d_silu(v[r]);
    }
    if (a.epi == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    if (a.epi == 5) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = l2d_gelu(v[r]);
    }
    if (resp) {
        h16x4 rr = *reinterpret_cast<const h16x4 *>(resp + (long long)m * a.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
    }
    if (a.epi == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    h16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (h16)v[r];
    *reinterpret_cast<h16x4 *>(outp + (long long)m * a.ldo + n) = o;
}

// MODE 0: linear / 1x1 (taps = 1);  MODE 1: 3x3, single input, no upsample (fast gather);  MODE 2: 3x3 generic
template <int TN, int TM, int MODE, int BK, int NS>
__global__ __launch_bounds__(256, (TN == 128 && BK == 32) ? (NS == 2 ? 4 : 3) : 1) void igemm_kernel(IGemmArgs a) {
    constexpr int NI = TN / 32;        // 16-row fragments per wave along channels
    constexpr int MI = TM / 32;        // 16-col fragments per wave along tokens
    constexpr int SPR = BK / 8;        // 16-byte slots per LDS row (4 or 8)
    constexpr int RPI = 64 / SPR;      // rows covered by one DMA wave-instruction (16 or 8)
    constexpr int NIW = TN / RPI / 4;  // DMA instructions per wave per stage (weights)
    constexpr int NIX = TM / RPI / 4;  // ... (tokens)
    constexpr int LPS = NIW + NIX;
    constexpr int KK = BK / 32;        // MFMA K sub-steps per stage
    constexpr int STAGE = (TN + TM) * BK;   // halfs
    extern __shared__ __attribute__((aligned(16))) h16 smem[];   // NS * STAGE halfs (the ONLY LDS object)

    // Touch every 64-byte line of the argument block NOW (four throw-away scalar loads, all in flight together): the
    // compiler fetches arguments where they are first used, which made the path to the first DMA a chain of 4-5 dependent
    // scalar-cache misses.  The destination registers stay reserved until the asm statement below the first DMA issue; the
    // compiler's own lgkmcnt(0) waits in between cover these loads too (the counter is per wave, not per instruction).
    unsigned ka0, ka1, ka2, ka3;
    {
        const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
        static_assert(sizeof(IGemmArgs) > 0xc0 + 4, "argument block shrank: adjust the warm-up offsets");
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0"
                     : "=&s"(ka0), "=&s"(ka1), "=&s"(ka2), "=&s"(ka3)
                     : "s"(kp));
    }
    L2D_STAMP(0);                                       // block entry
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    auto swz = [](int r) { return BK == 32 ? ((r >> 1) & 3) : (BK == 64 ? (r & 7) : (r & 15)); };

    // XCD-aware tile order (blocks are dispatched round-robin over the 8 XCDs): bijective remap so that
    // consecutive tiles -- same token tile, neighbouring weight tiles -- share one XCD's L2
    const int nwg = a.nwg;             // == gridDim.x (from the argument block: gridDim comes through one more scalar load)
    int wgid;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // Each XCD runs a contiguous range of wgid.  order 0: token-tile major (an XCD owns a band of token rows and walks
    // all weight tiles: every XCD's L2 pulls the whole weight matrix -- right when activations dominate); order 1:
    // weight-tile major (an XCD owns a band of output channels: the weights enter ONE L2, the small activation
    // matrix enters all eight -- right for the low-resolution levels, where weights are 10-100x the activations).
    int tile_n, tile_m;
    if (a.order) {
        tile_n = l2d_divf(wgid, a.ntm, a.inv_ntm);
        tile_m = wgid - tile_n * a.ntm;
    } else {
        tile_m = l2d_divf(wgid, a.ntn, a.inv_ntn);
        tile_n = wgid - tile_m * a.ntn;
    }
    const int n0 = tile_n * TN, m0 = tile_m * TM;
    const long long z = blockIdx.z;
    const h16 *wp = a.w + z * a.sw;

    // split-K: this block owns K steps [kb, ke)
    const int nk = a.Kp / BK;
    int kb = 0, ke = nk;
    if (a.splitk > 1) {                // nk * S < 2^24
        kb = l2d_divf(nk * (int)blockIdx.y, a.splitk, a.inv_s);
        ke = l2d_divf(nk * ((int)blockIdx.y + 1), a.splitk, a.inv_s);
    }

    // issue_w() / issue_x() are called for consecutive stages kb, kb+1, ...: ring slot, k offset and the conv tap are
    // tracked incrementally (no division / modulo in the loop); everything except the final add/select is wave-uniform
    int is_slot = 0, is_tap = 0, is_cb = kb * BK;
    if (MODE != 0) {
        if (kb > 0) {                  // (split launches only)
            is_tap = (kb * BK) / a.CinP;
            is_cb = kb * BK - is_tap * a.CinP;
        }
    }

    // ---- per-lane DMA descriptors.  Within an RPI-row group lane l serves row l/SPR, physical slot l%SPR.
    const int lrow = lane / SPR, pslot = lane % SPR;
    const h16 *wptr[NIW];      // advances by wadv (BK or 0) halfs per stage
    int wadv[NIW];
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const int r = (j * 4 + wave) * RPI + lrow;
        const int ls = pslot ^ swz(r);
        const int n = n0 + r;
        const bool ok = n < a.Nout;
        wptr[j] = ok ? wp + (long long)n * a.Kp + ls * 8 + (long long)kb * BK : a.zero;
        wadv[j] = ok ? BK : 0;
    }
    auto issue_w = [&]() {
        h16 *st = smem + is_slot * STAGE;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int grp = j * 4 + wave;
            __builtin_amdgcn_global_load_lds(L2D_GPTR(wptr[j]), L2D_LPTR(st + grp * RPI * BK), 16, 0, 0);
            wptr[j] += wadv[j];
        }
    };
    // The weight tile of the first stage needs nothing but the kernel arguments and the block id: its DMA leaves before
    // the token-row descriptors (integer divisions, 9-tap masks) are computed, so that arithmetic runs under the first
    // (cold: the previous kernel's end flushed the L2s) memory round trip instead of in front of it.
    if (kb < ke) issue_w();
    L2D_STAMP(1);                                       // first weight stage issued
    // end of the warm-up registers' reservation.  The explicit wait is free here (the first DMA needed the arguments, so the
    // compiler has already waited for every scalar load) and makes the "loads have landed before the registers are reused"
    // guarantee independent of the compiler's scheduling.
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(ka0), "s"(ka1), "s"(ka2), "s"(ka3));
    const h16 *x1 = a.x1 + z * a.sx1;
    h16 *outp = a.out + z * a.so;
    const h16 *resp = a.res ? a.res + z * a.sres : nullptr;
    const int Ctot = a.C1 + a.C2;

    // token rows: xoff = element offset of (row, channel slot) from x1 for tap (0,0) [MODE 1] / for k = 0 [MODE 0]
    long long xoff[NIX];
    int xls[NIX], xmask[NIX];          // xmask: bit t = tap t in bounds (MODE 1); bit 0 = row valid (MODE 0)
    int xb[NIX], xy[NIX], xx[NIX];     // MODE 2 only
#pragma unroll
    for (int j = 0; j < NIX; ++j) {
        const int r = (j * 4 + wave) * RPI + lrow;
        xls[j] = (pslot ^ swz(r)) * 8;
        const int m = m0 + r;
        const bool rv = m < a.M;
        xb[j] = xy[j] = xx[j] = 0;
        if (MODE == 0) {
            xoff[j] = (long long)m * a.ldx1 + xls[j];
            xmask[j] = rv ? 1 : 0;
            xb[j] = m;
        } else {
            const int hw = a.Hout * a.Wout;
            const int b = l2d_divf(m, hw, a.inv_hw), rr = m - b * hw;      // m < 2^24 (validated)
            const int oy = l2d_divf(rr, a.Wout, a.inv_wout), ox = rr - oy * a.Wout;
            xb[j] = b; xy[j] = oy; xx[j] = ox;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            xoff[j] = (((long long)b * a.Hin + iy0) * a.Win + ix0) * a.ldx1 + xls[j];
            int mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                if (rv && iy >= 0 && ix >= 0 && iy < a.Hin && ix < a.Win) mk |= 1 << t;
            }
            xmask[j] = mk;
        }
    }

    auto issue_x = [&]() {
        h16 *st = smem + is_slot * STAGE;
        if (MODE == 0) {
            const bool two = a.C2 > 0;
#pragma unroll
            for (int j = 0; j < NIX; ++j) {
                const int grp = j * 4 + wave;
                const int c = is_cb + xls[j];
                const h16 *src = a.zero;
                if (xmask[j]) {
                    if (c < a.C1) src = x1 + xoff[j] + is_cb;
                    else if (two && c < Ctot) src = a.x2 + (long long)xb[j] * a.ldx2 + (c - a.C1);
                }
                __builtin_amdgcn_global_load_lds(L2D_GPTR(src), L2D_LPTR(st + (TN + grp * RPI) * BK), 16, 0, 0);
            }
        } else if (MODE == 1) {
            const int ky = (is_tap * 11) >> 5;          // tap / 3 for tap in [0, 9)
            const int kx = is_tap - ky * 3;
            const long long uoff = (long long)(ky * a.Win + kx) * a.ldx1 + is_cb;   // wave-uniform
#pragma unroll
            for (int j = 0; j < NIX; ++j) {
                const int grp = j * 4 + wave;
                const bool ok = ((xmask[j] >> is_tap) & 1) && (is_cb + xls[j] < a.C1);
                const h16 *src = ok ? x1 + xoff[j] + uoff : a.zero;
                __builtin_amdgcn_global_load_lds(L2D_GPTR(src), L2D_LPTR(st + (TN + grp * RPI) * BK), 16, 0, 0);
            }
        } else {
            const int ky = (is_tap * 11) >> 5;
            const int kx = is_tap - ky * 3;
#pragma unroll
            for (int j = 0; j < NIX; ++j) {
                const int grp = j * 4 + wave;
                const int c = is_cb + xls[j];
                const int iy = xy[j] * a.stride + ky - a.pad;
                const int ix = xx[j] * a.stride + kx - a.pad;
                const bool ok = (m0 + (j * 4 + wave) * RPI + lrow < a.M) && c < Ctot && iy >= 0 && ix >= 0 &&
                                iy < (a.Hin << a.ups) && ix < (a.Win << a.ups);
                const long long pix = ((long long)xb[j] * a.Hin + (iy >> a.ups)) * a.Win + (ix >> a.ups);
                const h16 *src = a.zero;
                if (ok) src = (c < a.C1) ? x1 + pix * a.ldx1 + c : a.x2 + pix * a.ldx2 + (c - a.C1);
                __builtin_amdgcn_global_load_lds(L2D_GPTR(src), L2D_LPTR(st + (TN + grp * RPI) * BK), 16, 0, 0);
            }
        }
        is_slot = (is_slot + 1 == NS) ? 0 : is_slot + 1;
        is_cb += BK;
        if (MODE != 0 && is_cb >= a.CinP) { is_cb -= a.CinP; ++is_tap; }
    };
    auto issue = [&]() { issue_w(); issue_x(); };

    // ---- epilogue operands that do not depend on the accumulators are fetched NOW (plain loads, older than every token
    // DMA in the in-order VMEM queue: the loop's counted waits cover them), not after the K loop where each would be one
    // more dependent (cold) round trip on the block's critical path.
    const int NoutO = (a.epi == 1) ? (a.Nout >> 1) : a.Nout;            // GEGLU halves the output width
    const bool split = a.splitk > 1;
    const bool vec = a.epl && (!split || a.cnt) && ((a.ldo | NoutO) & 7) == 0 && (((unsigned long long)outp) & 15) == 0 &&
                     (!resp || ((a.ldr & 7) == 0 && (((unsigned long long)resp) & 15) == 0));
    constexpr bool RES_EARLY = (TN * TM <= 64 * 64);   // 8 + 8 VGPRs; the 128x128 tile would need 32 + 16 across the loop and
                                                       // drop from 3 to 2 blocks per CU: it loads them when the loop ends
    constexpr int CPRN = TN / 8;                                        // 16-byte chunks per output row (non-GEGLU)
    constexpr int EPI_IT = (TM * CPRN) / 256;
    f32x4 biasv[NI];
    f32x4 rbv[RES_EARLY ? NI : 1];     // 64x64 tile: the row bias stays in its own registers until the epilogue -- adding it
                                       // here would make the prologue wait (vmcnt 0) for both loads behind the first DMA
    h16x8 resv[RES_EARLY ? EPI_IT : 1];
    bool rb_in_bias = false;
    auto load_bias = [&]() {
        const float *rbp = nullptr;
        if (a.rowbias) {
            const int mlast = (m0 + TM <= a.M ? m0 + TM : a.M) - 1;
            const int smp = l2d_divf(m0, a.rows_per_bias, 1.0f / (float)a.rows_per_bias);
            if (mlast < (smp + 1) * a.rows_per_bias) {                  // the whole tile lies in one sample
                rbp = a.rowbias + (long long)smp * a.ldrb;
                rb_in_bias = true;
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + wn * (TN / 2) + i * 16 + lg * 4;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (n < a.Nout) {
                if (a.bias) b = *reinterpret_cast<const f32x4 *>(a.bias + n);
                if (rbp) {
                    if constexpr (RES_EARLY) rbv[i] = *reinterpret_cast<const f32x4 *>(rbp + n);
                    else b += *reinterpret_cast<const f32x4 *>(rbp + n);
                }
            }
            if constexpr (RES_EARLY) { if (!rbp || n >= a.Nout) rbv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            biasv[i] = b;
        }
    };
    auto load_early = [&]() {
        load_bias();
        if (resp) {
#pragma unroll
            for (int it = 0; it < EPI_IT; ++it) {
                const int c = it * 256 + tid, row = c / CPRN, cc = c % CPRN;
                const int m = m0 + row, n = n0 + cc * 8;
                resv[RES_EARLY ? it : 0] = (m < a.M && n < a.Nout) ? l2d_ld8(resp + (long long)m * a.ldr + n) : l2d_zero8();
            }
        }
    };
    if (vec && RES_EARLY && !split) load_early();   // (a split launch fetches them in the one block that runs the epilogue)
    if (kb < ke) issue_x();

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets are loop invariant (halfs, relative to the stage base)
    int aoff[KK][NI], boff[KK][MI];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = wn * (TN / 2) + i * 16 + li;
            aoff[kk][i] = r * BK + (((kk * 4 + lg) ^ swz(r)) << 3);
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int r = wm * (TM / 2) + j * 16 + li;
            boff[kk][j] = (TN + r) * BK + (((kk * 4 + lg) ^ swz(r)) << 3);
        }
    }
    int cp_slot = 0;
    auto compute = [&]() {
        const h16 *st = smem + cp_slot * STAGE;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            h16x8 af[NI], bf[MI];
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i] = l2d_ld8(st + aoff[kk][i]);
#pragma unroll
            for (int j = 0; j < MI; ++j) bf[j] = l2d_ld8(st + boff[kk][j]);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        cp_slot = (cp_slot + 1 == NS) ? 0 : cp_slot + 1;
    };

    // prologue: NS-1 stages in flight (stage kb went out above)
#pragma unroll
    for (int s = 1; s < NS - 1; ++s)
        if (kb + s < ke) issue();
    L2D_STAMP(2);                                       // descriptors computed, prologue stages issued
    // steady state: stage kt has landed when at most (NS-2) younger stages are still outstanding
    int kt = kb;
    for (; kt + (NS - 1) < ke; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
        __builtin_amdgcn_s_barrier();      // every wave's share of stage kt is in LDS; everyone finished stage kt-1
        if (kt == kb) L2D_STAMP(3);        // first stage landed
        issue();                           // refill the ring slot that stage kt-1 occupied
        compute();
    }
    // drain: no more refills; wait for everything, then the remaining (<= NS-1) stages
    for (; kt < ke; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        compute();
    }

    // ---------------------------------------------------------------- epilogue
    L2D_STAMP(4);                                       // K loop done
    if (split && a.cnt) {
        // split-K with the reduction fused into this launch: every block parks its fp32 partial tile in the workspace
        // (tile-private slab, lane-linear: 16 bytes per lane, read back by the same thread positions), then announces itself
        // on the tile's arrival counter; the block that arrives LAST sums the S partials in the fixed order 0..S-1 -- its own
        // included, so the result does not depend on who was last: bit-repeatable -- and runs the normal fused epilogue
        // (bias / activation / residual / GroupNorm statistics).  No second launch, no second pass over the output.
        // Cross-XCD visibility WITHOUT cache-wide fences: the partials are written and read with agent-scope (sc1) buffer
        // accesses -- write-through stores, coherent loads -- and ordered against the arrival atomic by s_waitcnt + the block
        // barrier.  (__threadfence() here = buffer_wbl2 + buffer_inv of the whole L2 per block: measured 42 us per launch,
        // the frame went from 9.9 to 12.1 ms.)
        const int S = a.splitk;
        constexpr int AUX_SC1 = 16;
        float *slab = a.ws + ((z * nwg + wgid) * S) * (long long)(TN * TM);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, S * TN * TM * 4, 0x00020000);
        const int mine = ((int)blockIdx.y * (TN * TM) + tid * 4) * 4;          // byte offsets inside the tile's slab
        // (fragments whose 16 token rows all lie beyond M -- most of a tile at the 8x8 level -- are neither parked nor summed:
        // wave-uniform test, the same on both sides)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            if (m0 + wm * (TM / 2) + j * 16 >= a.M) continue;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, mine + (i * MI + j) * 4096, 0, AUX_SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's partials have been written through ...
        __syncthreads();                                    // ... every thread's have
        unsigned int *flag = reinterpret_cast<unsigned int *>(smem);
        if (tid == 0) *flag = atomicAdd(a.cnt + (z * nwg + wgid), 1u);
        __syncthreads();
        const bool last = (*flag == (unsigned int)(S - 1));
        __syncthreads();                                    // (smem is reused below)
        if (!last) return;
        if (tid == 0) atomicExch(a.cnt + (z * nwg + wgid), 0u);                // ready for the next launch that uses this counter
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            if (m0 + wm * (TM / 2) + j * 16 >= a.M) continue;
#pragma unroll 4
            for (int y = 0; y < S; ++y) {                   // (unrolled: four slabs of loads in flight; the sum keeps its order)
                const int src = (y * (TN * TM) + tid * 4) * 4;
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    acc[i][j] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, src + (i * MI + j) * 4096, 0, AUX_SC1));
            }
        }
        if (vec && RES_EARLY) load_early();
    } else if (split) {
        // split-K, two launches: raw fp32 partial tile; the fused epilogue runs in igemm_splitk_epilogue
        const int NoutP = (a.Nout + 3) & ~3;
        float *wsp = a.ws + ((long long)z * a.splitk + blockIdx.y) * a.M * NoutP;
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int m = m0 + wm * (TM / 2) + j * 16 + li;
            if (m >= a.M) continue;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int n = n0 + wn * (TN / 2) + i * 16 + lg * 4;
                if (n >= a.Nout) continue;
                *reinterpret_cast<f32x4 *>(wsp + (long long)m * NoutP + n) = acc[i][j];
            }
        }
        return;
    }
    if (vec) {
        // ---- epilogue through LDS: the accumulator layout gives a lane 4 channels of one token (8-byte pieces, 16 rows
        // of 32 bytes per store instruction); the finished fp16 tile is therefore transposed through the (now idle) ring
        // into [token][channel] rows and written back as whole rows with 16 bytes per lane, residual added on the way
        // with 16-byte loads.  Rounding points are those of the reference's fp16 graph: conv / linear output (bias,
        // activation in fp32) -> fp16, then the residual add in fp16.
        const int pitch = (a.epi == 1 ? TN / 2 : TN) + 8;               // halfs; row stride = 16 B mod 32 B
        h16 *ot = smem;
        if (!RES_EARLY) load_bias();
        __syncthreads();                                                // every wave is done reading the last stage
        if (a.epi == 1) {
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int row = wm * (TM / 2) + j * 16 + li;
#pragma unroll
                for (int p = 0; p < NI / 2; ++p) {
                    const int col = wn * (TN / 4) + p * 16 + lg * 4;
                    h16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[2 * p][j][r] + biasv[2 * p][r];
                        const float g = acc[2 * p + 1][j][r] + biasv[2 * p + 1][r];
                        o[r] = (h16)(v * l2d_gelu(g));
                    }
                    *reinterpret_cast<h16x4 *>(ot + row * pitch + col) = o;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int row = wm * (TM / 2) + j * 16 + li;
                const float *rb = (a.rowbias && !rb_in_bias)
                                      ? a.rowbias + (long long)((m0 + row < a.M ? m0 + row : a.M - 1) / a.rows_per_bias) * a.ldrb
                                      : nullptr;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int col = wn * (TN / 2) + i * 16 + lg * 4;
                    f32x4 v = acc[i][j] + biasv[i];
                    if constexpr (RES_EARLY) v += rbv[i];
                    if (rb && n0 + col < a.Nout) v += *reinterpret_cast<const f32x4 *>(rb + n0 + col);
                    if (a.epi == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = l2d_silu(v[r]);
                    }
                    if (a.epi == 3) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (a.epi == 5) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = l2d_gelu(v[r]);
                    }
                    h16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (h16)v[r];
                    *reinterpret_cast<h16x4 *>(ot + row * pitch + col) = o;
                }
            }
        }
        constexpr int CPRN_LOG = (CPRN == 16) ? 4 : 3;
        const int cshift = (a.epi == 1) ? CPRN_LOG - 1 : CPRN_LOG;      // 16-byte chunks per output row = 1 << cshift
        const int n0o = (a.epi == 1) ? (n0 >> 1) : n0;
        L2D_STAMP(5);                                   // tile transposed into LDS (issue side)
        h16x8 rlate[RES_EARLY ? 1 : EPI_IT];
        if (!RES_EARLY && resp) {                                       // big tile: residual loads fly during the transpose
#pragma unroll
            for (int it = 0; it < EPI_IT; ++it) {
                const int c = it * 256 + tid, row = c / CPRN, cc = c % CPRN;
                const int m = m0 + row, n = n0 + cc * 8;
                rlate[it] = (m < a.M && n < a.Nout) ? l2d_ld8(resp + (long long)m * a.ldr + n) : l2d_zero8();
            }
        }
        __syncthreads();
        const bool gn = a.gn1 != nullptr;
        // this thread's 8 channels over its rows, as 4 channel PAIRS: every group size here is even (C is a multiple of 64,
        // G = 32), so a pair never straddles two groups and v_dot2_f32_f16 does two channels per instruction
        float gs[4], gq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
        const h16x2 ones2 = {(h16)1.0f, (h16)1.0f};
#pragma unroll
        for (int it = 0; it < EPI_IT; ++it) {
            const int c = it * 256 + tid;
            const int row = c >> cshift, cc = c & ((1 << cshift) - 1);
            if (row >= TM) break;                                       // GEGLU tiles have half the chunks
            const int m = m0 + row, n = n0o + cc * 8;
            if (m >= a.M || n >= NoutO) continue;
            h16x8 v = l2d_ld8(ot + row * pitch + cc * 8);
            if (resp) v = v + (RES_EARLY ? resv[RES_EARLY ? it : 0] : rlate[RES_EARLY ? 0 : it]);
            if (a.epi == 4) v = __builtin_elementwise_max(v, l2d_zero8());          // relu(conv + skip), TAESD blocks
            l2d_st8(outp + (long long)m * a.ldo + n, v);
            if (gn) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const h16x2 pr = {v[2 * e], v[2 * e + 1]};
                    gs[e] = __builtin_amdgcn_fdot2(pr, ones2, gs[e], false);
                    gq[e] = __builtin_amdgcn_fdot2(pr, pr, gq[e], false);
                }
            }
        }
        L2D_STAMP(6);                                   // row stores issued
        if (gn) {
            // GroupNorm statistics of what was just stored (the fp16 values the consumer will read): thread -> channel
            // -> group inside the block, then two integer atomics per (consumer, overlapped group).  The plan builder only
            // asks for this when a tile lies inside one sample.
            __syncthreads();                                            // every thread is done reading the staged tile
            float *red = reinterpret_cast<float *>(smem);               // [256][8]: 4 pair sums | 4 pair sums of squares
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = gs[e]; red[tid * 8 + 4 + e] = gq[e]; }
            __syncthreads();
            float *chs1 = red + 256 * 8, *chs2 = chs1 + TN / 2;         // per channel pair
            const int tno = (a.epi == 1) ? TN / 2 : TN;
            if (tid < tno / 2) {
                const int cc = tid >> 2, e = tid & 3;
                float s = 0.f, q = 0.f;
                for (int r = 0; r < (256 >> cshift); ++r) { s += red[((r << cshift) | cc) * 8 + e]; q += red[((r << cshift) | cc) * 8 + 4 + e]; }
                chs1[tid] = s; chs2[tid] = q;
            }
            __syncthreads();
            // group sums in units of channel pairs (cpg, offsets and tile origin are all even)
            const int nch = min(tno, NoutO - n0o), bsmp = m0 / a.gnT;
            l2d_gn_flush(a.gn1, a.gnG, a.cpg1 >> 1, a.choff1 >> 1, bsmp, chs1, chs2, n0o >> 1, nch >> 1, tid);
            l2d_gn_flush(a.gn2, a.gnG, a.cpg2 >> 1, a.choff2 >> 1, bsmp, chs1, chs2, n0o >> 1, nch >> 1, tid);
        }
        L2D_STAMP(7);                                   // block done (issue side)
        return;
    }
    if (a.epi == 1) {
        // GEGLU: fragment pairs (2p, 2p+1) hold value / gate of the same output columns
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int m = m0 + wm * (TM / 2) + j * 16 + li;
            if (m >= a.M) continue;
#pragma unroll
            for (int p = 0; p < NI / 2; ++p) {
                const int nv = n0 + wn * (TN / 2) + (2 * p) * 16 + lg * 4;      // packed row of the value part
                const int ng = nv + 16;                                         // packed row of the gate part
                if (ng >= a.Nout) continue;
                const int no = (n0 + wn * (TN / 2)) / 2 + p * 16 + lg * 4;      // output column
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.bias + nv);
                const f32x4 bg = *reinterpret_cast<const f32x4 *>(a.bias + ng);
                h16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[2 * p][j][r] + bv[r];
                    const float g = acc[2 * p + 1][j][r] + bg[r];
                    o[r] = (h16)(v * l2d_gelu(g));
                }
                *reinterpret_cast<h16x4 *>(outp + (long long)m * a.ldo + no) = o;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int m = m0 + wm * (TM / 2) + j * 16 + li;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + wn * (TN / 2) + i * 16 + lg * 4;
            if (n >= a.Nout) continue;
            igemm_epilogue(a, outp, resp, m, n, acc[i][j]);
        }
    }
}

// sums the S fp32 partial tiles of a split-K launch and applies the fused epilogue.  A block owns a 64-token x 64-channel
// tile (thread = 4 channels x 4 token rows), which also lets it accumulate the GroupNorm statistics of its output.
__global__ __launch_bounds__(256) void igemm_splitk_epilogue(IGemmArgs a, int S) {
    const int NoutP = (a.Nout + 3) & ~3;
    const int tiles_n = (NoutP + 63) / 64;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int tid = threadIdx.x, cq = tid & 15, r0 = tid >> 4;
    const int n = tn * 64 + cq * 4;
    const long long z = blockIdx.z;
    const long long slab = (long long)a.M * NoutP;
    const bool gn = a.gn1 != nullptr;
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < NoutP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = tm * 64 + r0 + 16 * k;
            if (m >= a.M) break;
            const float *wsp = a.ws + (long long)z * S * slab + (long long)m * NoutP + n;
            f32x4 v = *reinterpret_cast<const f32x4 *>(wsp);
#pragma unroll 8
            for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4 *>(wsp + s * slab);   // (unrolled: loads in flight together; same order)
            h16 *outp = a.out + z * a.so;
            igemm_epilogue(a, outp, a.res ? a.res + z * a.sres : nullptr, m, n, v);
            if (gn) {                       // what was stored, as the consumer will read it (same thread, same addresses)
                const h16x4 o = *reinterpret_cast<const h16x4 *>(outp + (long long)m * a.ldo + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float f = (float)o[e]; gs[e] += f; gq[e] = fmaf(f, f, gq[e]); }
            }
        }
    }
    if (gn) {
        __shared__ float red[256][8];
        __shared__ float chs[2][64];
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[tid][e] = gs[e]; red[tid][4 + e] = gq[e]; }
        __syncthreads();
        if (tid < 64) {
            const int c4 = tid >> 2, e = tid & 3;
            float s = 0.f, q = 0.f;
            for (int r = 0; r < 16; ++r) { s += red[r * 16 + c4][e]; q += red[r * 16 + c4][4 + e]; }
            chs[0][tid] = s; chs[1][tid] = q;
        }
        __syncthreads();
        const int nch = min(64, a.Nout - tn * 64), bsmp = (tm * 64) / a.gnT;
        l2d_gn_flush(a.gn1, a.gnG, a.cpg1, a.choff1, bsmp, chs[0], chs[1], tn * 64, nch, tid);
        l2d_gn_flush(a.gn2, a.gnG, a.cpg2, a.choff2, bsmp, chs[0], chs[1], tn * 64, nch, tid);
    }
}

template <int TN, int TM, int MODE, int BK, int NS>
static void launch_v(const IGemmArgs &a, int batch, hipStream_t s) {
    constexpr size_t RING = (size_t)NS * (TN + TM) * BK * sizeof(h16);
    constexpr size_t EPI = (size_t)TM * (TN + 8) * sizeof(h16);        // the LDS-staged epilogue's transposed tile
    constexpr size_t LDS = RING > EPI ? RING : EPI;
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (LDS > 65536 && !attr_done) {   // > 64 KB of dynamic LDS must be opted into once per kernel
        // (fails inside a stream capture: the plan's first run is always direct -- HipStreamingUNet._run; if it fails
        // anyway the flag stays clear, the launch below reports the error and the next direct run retries)
        if (hipFuncSetAttribute((const void *)igemm_kernel<TN, TM, MODE, BK, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) == hipSuccess)
            attr_done = true;
        else
            (void)hipGetLastError();
    }
    int ntn = (a.Nout + TN - 1) / TN, ntm = (a.M + TM - 1) / TM;
    IGemmArgs b = a;
    b.ntn = ntn; b.ntm = ntm; b.nwg = ntn * ntm;
    b.inv_ntn = 1.0f / (float)ntn; b.inv_ntm = 1.0f / (float)ntm; b.inv_s = 1.0f / (float)a.splitk;
    b.inv_hw = 1.0f / (float)(a.Hout * a.Wout > 0 ? a.Hout * a.Wout : 1); b.inv_wout = 1.0f / (float)(a.Wout > 0 ? a.Wout : 1);
    dim3 grid(ntn * ntm, a.splitk, batch), block(256);
    hipLaunchKernelGGL((igemm_kernel<TN, TM, MODE, BK, NS>), grid, block, LDS, s, b);
}

// pipeline variants (op.i[23]): 0 = BK32 x 4 stages, 1 = BK64 x 3, 2 = BK64 x 4, 3 = BK32 x 6, 4 = BK32 x 3,
// 5 = BK64 x 2, 6 = BK128 x 2, 7 = BK128 x 3 (64x64 tile only), 8 = BK64 x 6 (64x64 only), 9 = BK64 x 4 (64x64 only),
// 10 = BK32 x 2
template <int TN, int TM, int MODE>
static int launch_p(const IGemmArgs &a, int batch, int variant, hipStream_t s) {
    switch (variant) {
        case 0: launch_v<TN, TM, MODE, 32, 4>(a, batch, s); return L2D_OK;
        case 1: launch_v<TN, TM, MODE, 64, 3>(a, batch, s); return L2D_OK;
        case 2: launch_v<TN, TM, MODE, 64, 4>(a, batch, s); return L2D_OK;
        case 3: launch_v<TN, TM, MODE, 32, 6>(a, batch, s); return L2D_OK;
        case 4: launch_v<TN, TM, MODE, 32, 3>(a, batch, s); return L2D_OK;
        case 5: launch_v<TN, TM, MODE, 64, 2>(a, batch, s); return L2D_OK;
        case 6: launch_v<TN, TM, MODE, 128, 2>(a, batch, s); return L2D_OK;
        case 7: if constexpr (TN + TM <= 128) { launch_v<TN, TM, MODE, 128, 3>(a, batch, s); return L2D_OK; } break;
        case 8: if constexpr (TN + TM <= 128) { launch_v<TN, TM, MODE, 64, 6>(a, batch, s); return L2D_OK; } break;
        case 9: if constexpr (TN + TM <= 128) { launch_v<TN, TM, MODE, 64, 4>(a, batch, s); return L2D_OK; } break;
        case 10: launch_v<TN, TM, MODE, 32, 2>(a, batch, s); return L2D_OK;   // shallow ring: 4 blocks of the 128x128 tile per CU
    }
    return L2D_EINVAL;
}

template <int TN, int TM>
static int launch_t(const IGemmArgs &a, int batch, int variant, hipStream_t s) {
    if (a.taps == 1) return launch_p<TN, TM, 0>(a, batch, variant, s);
    if (a.C2 == 0 && a.ups == 0) return launch_p<TN, TM, 1>(a, batch, variant, s);
    return launch_p<TN, TM, 2>(a, batch, variant, s);
}

int l2d_launch_igemm(const l2d_op *op, hipStream_t s) {
    IGemmArgs a;
    a.x1 = (const h16 *)op->p[0]; a.x2 = (const h16 *)op->p[1]; a.w = (const h16 *)op->p[2];
    a.bias = (const float *)op->p[3]; a.rowbias = (const float *)op->p[4];
    a.res = (const h16 *)op->p[5]; a.out = (h16 *)op->p[6]; a.zero = (const h16 *)op->p[7]; a.ws = (float *)op->p[8];
    a.taps = op->i[0]; a.C1 = op->i[1]; a.C2 = op->i[2]; a.ldx1 = op->i[3]; a.ldx2 = op->i[4];
    a.CinP = op->i[5]; a.B = op->i[6]; a.Hin = op->i[7]; a.Win = op->i[8]; a.Hout = op->i[9];
    a.Wout = op->i[10]; a.stride = op->i[11]; a.ups = op->i[12]; a.M = op->i[13]; a.Nout = op->i[14];
    a.ldo = op->i[15]; a.ldr = op->i[16]; a.ldrb = op->i[17]; a.rows_per_bias = op->i[18]; a.epi = op->i[19];
    int batch = op->i[20] > 0 ? op->i[20] : 1;
    a.splitk = op->i[21] > 0 ? op->i[21] : 1;
    int tile = op->i[22] & 15; // 0 auto, 1 = 128x128, 2 = 64x64
    a.order = (op->i[22] >> 4) & 1;   // XCD tile order: 0 token-tile major, 1 weight-tile major
    a.epl = ((op->i[22] >> 5) & 1) ? 0 : 1;   // + 32: direct (register -> global, 8-byte pieces) epilogue instead of the LDS-staged one
    int variant = op->i[23];   // pipeline variant, see launch_p
    a.sx1 = op->l[0]; a.sw = op->l[1]; a.so = op->l[2]; a.sres = op->l[3];
    a.gn1 = (unsigned long long *)op->p[9]; a.gn2 = (unsigned long long *)op->p[10];
    a.cnt = (unsigned int *)op->p[11];
#ifdef L2D_PROBES
    a.probe = g_igemm_probe;
#endif
    a.pad = op->i[30] ? 0 : 1;        // i30 = 1: TF-"SAME" low-side padding 0 (stride-2 convs of the ResNetV2 backbone)
    a.gnT = op->i[24]; a.gnG = op->i[25]; a.cpg1 = op->i[26]; a.choff1 = op->i[27]; a.cpg2 = op->i[28]; a.choff2 = op->i[29];
    if (!a.gn1 && a.gn2) { a.gn1 = a.gn2; a.cpg1 = a.cpg2; a.choff1 = a.choff2; a.gn2 = nullptr; }
    a.Kp = a.taps * a.CinP;
    if (!a.x1 || !a.w || !a.out || !a.zero || (a.taps != 1 && a.taps != 9) || a.CinP <= 0 || a.CinP % 64 != 0 ||
        a.M <= 0 || a.Nout <= 0 || (a.C1 % 8) || (a.C2 % 8) || (a.C2 > 0 && !a.x2) || a.C1 + a.C2 > a.CinP ||
        (a.ldo % 4) || (a.ldx1 % 8) || (a.C2 > 0 && (a.ldx2 % 8)) ||
        (a.res && (a.ldr % 4)) || (a.rowbias && a.rows_per_bias <= 0) ||
        (a.epi == 1 && (!a.bias || (a.Nout % 32) || a.splitk != 1 || a.res || a.rowbias)) || (a.stride != 1 && a.stride != 2) ||
        (a.cnt && a.splitk > 1 && tile == 0) || a.epi < 0 || a.epi > 5 || (a.ups != 0 && a.ups != 1) || tile < 0 || tile > 2 || (a.splitk > 1 && !a.ws) || a.splitk > a.Kp / 64 ||
        variant < 0 || variant > 10 || a.splitk > 64 || (a.CinP % 128 != 0 && (variant == 6 || variant == 7))) {
        l2d_set_error("igemm(tag %d): invalid arguments (taps=%d C1=%d C2=%d CinP=%d M=%d Nout=%d ldo=%d splitk=%d tile=%d zero=%p)",
                      op->tag, a.taps, a.C1, a.C2, a.CinP, a.M, a.Nout, a.ldo, a.splitk, tile, (const void *)a.zero);
        return L2D_EINVAL;
    }
    if (a.gn1) {
        // (tile 0 = auto is resolved below: require the larger one; the two-launch split-K reduction works on 64x64 tiles)
        const int tm = ((a.splitk > 1 && !a.cnt) || tile == 2) ? 64 : 128;
        // (the kernel's LDS-staged epilogue -- the only one that accumulates statistics -- also needs 16-byte aligned out /
        // residual pointers: a misaligned sub-view would silently take the direct epilogue and leave the accumulators zero)
        const bool vec_ok = a.epl && (a.Nout % 8) == 0 && (a.ldo % 8) == 0 && !(a.res && (a.ldr % 8)) && a.epi != 1 &&
                            (((unsigned long long)a.out) & 15) == 0 && (((unsigned long long)a.res) & 15) == 0;
        if (a.gnT <= 0 || a.gnG <= 0 || a.gnG > 32 || a.cpg1 <= 0 || (a.gn2 && a.cpg2 <= 0) || batch != 1 ||
            ((a.cpg1 | a.choff1) & 1) || (a.gn2 && ((a.cpg2 | a.choff2) & 1)) ||
            (a.gnT % tm) != 0 || ((a.splitk == 1 || a.cnt) && !vec_ok) || (a.M % a.gnT) != 0) {
            l2d_set_error("igemm(tag %d): GroupNorm statistics need T %% tile == 0 (T=%d tile=%d), the LDS-staged epilogue or "
                          "split-K, batch 1 and G <= 32", op->tag, a.gnT, tm);
            return L2D_EINVAL;
        }
    }
    if (a.M >= (1 << 24) || (long long)((a.M + 63) / 64) * ((a.Nout + 63) / 64) >= (1 << 24)) {
        l2d_set_error("igemm(tag %d): M = %d / tile count beyond the 2^24 range of the kernel's index arithmetic", op->tag, a.M);
        return L2D_EINVAL;
    }
    if (a.taps == 9 && (a.M != a.B * a.Hout * a.Wout)) {
        l2d_set_error("igemm(tag %d): M != B*Hout*Wout", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    if (tile == 0) {
        long long big = (long long)((a.Nout + 127) / 128) * ((a.M + 127) / 128) * batch * a.splitk;
        tile = (big >= 384) ? 1 : 2;
    }
    int lrc = (tile == 1) ? launch_t<128, 128>(a, batch, variant, s) : launch_t<64, 64>(a, batch, variant, s);
    if (lrc != L2D_OK) {
        l2d_set_error("igemm(tag %d): pipeline variant %d not available for this tile", op->tag, variant);
        return lrc;
    }
    int rc = l2d_check_launch("igemm", op->tag);
    if (rc != L2D_OK || a.splitk == 1 || a.cnt) return rc;
    const int NoutP = (a.Nout + 3) & ~3;
    const unsigned tiles = (unsigned)(((NoutP + 63) / 64) * ((a.M + 63) / 64));
    hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(tiles, 1, batch), dim3(256), 0, s, a, a.splitk);
    return l2d_check_launch("igemm_splitk_epilogue", op->tag);
}
