// 3x3 stride-1 convolution on MFMA with the activation PATCH resident in LDS (gfx950).  Replaces the InflatedConv3d 3x3 convs
// of the resnet blocks (reference resnet.py:57-65,194,214,229-259) at the resolutions where the implicit-GEMM kernel
// (igemm.hip) is bound by what a CU can ingest, not by the matrix cores.
//
// Why.  igemm.hip treats the conv as a GEMM over k = (tap, channel) and gathers a [tokens x 64] activation tile per K step:
// every activation is fetched 9 times (once per tap) by every one of the N / 64 channel-tile blocks of its token tile.  At
// the 64 x 64 level (M = 8192, N = 320, K = 2880) a launch moves 472 MB from L2 into the CUs for 10 MB of operands and runs at
// the ~11-13 TB/s the CUs ingest with LDS-DMA (profiles/r3z_frame_trace.csv: 35 us, 430 TFLOP/s), i.e. 32 flop per ingested
// byte.  Here a block owns a PH x PW patch of output pixels and 64 output channels:
//   * the input patch with its one-pixel halo ((PH+2) x (PW+2) pixels) is DMA'd into LDS ONCE per 64-channel chunk
//     (double-buffered) and serves all nine taps -- a tap is just a constant offset on the fragment's LDS address;
//     out-of-image pixels (the conv's zero padding) and the channel concat of two inputs are resolved per lane when the DMA
//     descriptors are built (zero page / second pointer), never in the loop;
//   * only the 64 x 64 weight tile of (tap, chunk) streams per stage (8 KB: two DMA instructions per wave) through a 3-stage
//     ring with counted vmcnt waits; the next chunk's patch rides along, one DMA instruction per wave and stage;
//   * k order = chunk-major, tap-minor (any order is the same sum); weights keep igemm's packing [Nout][tap][CinP].
// 8 x 16 patch, C = 320: 97 flop per ingested byte instead of 32; the MFMA work per stage is unchanged.
// Epilogue = igemm's LDS-staged one (bias, per-sample time-embedding bias, residual in fp16 after the fp16 rounding of the
// conv output, whole-row 16-byte stores, GroupNorm statistics of the output as fixed-point integer atomics).
#include <type_traits>

#include "common.h"

#define L2D_GPTR(p) ((__attribute__((address_space(1))) const void *)(p))
#define L2D_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

struct PConvArgs {
    const h16 *x1, *x2, *w, *zero;
    const float *bias, *rowbias;
    const h16 *res;
    h16 *out;
    int B, H, W, C1, C2, ldx1, ldx2, CinP, Nout, ldo, ldr, ldrb, rows_per_bias;
    int epi;                       // as igemm.hip: 0 none, 2 SiLU, 3 ReLU, 5 GELU (fp32, before the fp16 rounding), 4 ReLU after the residual add
    int npx, npy, ntn, nwg, order;
    unsigned long long *gn1, *gn2;
    int gnT, gnG, cpg1, choff1, cpg2, choff2;
};

template <int PH, int PW>
__global__ __launch_bounds__(256, 2) void pconv_kernel(PConvArgs a) {
    constexpr int TM = PH * PW, TN = 64, BK = 64, NSW = 3;
    constexpr int MI = TM / 32;                        // 16-token fragments per wave (2 x 2 waves: 32 channels x TM / 2 tokens)
    constexpr int PWH = PW + 2, NPIX = (PH + 2) * PWH;
    constexpr int NPIXP = (NPIX + 63) / 64 * 64;       // padded: a DMA instruction covers 64 pixels of one 16-byte channel slot
    constexpr int IPS = NPIXP / 64;                    // DMA instructions per channel slot
    constexpr int XI = 8 * IPS / 4;                    // patch DMA instructions per wave and chunk (8 slots of 8 channels)
    static_assert(XI <= 6 && TM % 32 == 0, "patch geometry");
    constexpr int PBUF = 8 * NPIXP * 8;                // halfs per patch buffer
    constexpr int WST = TN * BK;                       // halfs per weight stage
    extern __shared__ __attribute__((aligned(16))) h16 smem[];     // patch[2] | weight ring[NSW]   (the ONLY LDS object)
    h16 *pb0 = smem, *ring = smem + 2 * PBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1, li = lane & 15, lg = lane >> 4;

    // XCD-aware bijective block order (as igemm.hip): order 1 = weight-tile major
    int wgid;
    {
        const int q = a.nwg >> 3, r = a.nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int npat = a.npx * a.npy * a.B;
    int pat, tile_n;
    if (a.order) { tile_n = wgid / npat; pat = wgid - tile_n * npat; }
    else { pat = wgid / a.ntn; tile_n = wgid - pat * a.ntn; }
    const int n0 = tile_n * TN;
    const int bb = pat / (a.npx * a.npy), pr = pat - bb * (a.npx * a.npy);
    const int y0 = (pr / a.npx) * PH, x0 = (pr % a.npx) * PW;
    const int Ctot = a.C1 + a.C2, NCH = a.CinP / 64, Kp = 9 * a.CinP;
    const int total = 9 * NCH;

    // ---- weight DMA descriptors: instruction j of a stage moves rows (j * 4 + wave) * 8 + lane / 8, 16-byte slot lane % 8 ^ (row & 7)
    const h16 *wptr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (j * 4 + wave) * 8 + (lane >> 3);
        wptr[j] = a.w + (long long)(n0 + r) * Kp + (((lane & 7) ^ (r & 7)) << 3);
    }
    int w_slot = 0, w_tap = 0, w_cb = 0;               // next weight stage to request: ring slot, tap, channel base
    auto issue_w = [&]() {
        h16 *st = ring + w_slot * WST;
        const int kcol = w_tap * a.CinP + w_cb;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(L2D_GPTR(wptr[j] + kcol), L2D_LPTR(st + (j * 4 + wave) * 8 * BK), 16, 0, 0);
        w_slot = (w_slot + 1 == NSW) ? 0 : w_slot + 1;
        if (++w_tap == 9) { w_tap = 0; w_cb += 64; }
    };
    // ---- patch DMA descriptors: this lane's pixel of segment seg (pixel index seg * 64 + lane of the haloed patch)
    long long poff[IPS];                               // pixel index in the image (-1: outside the image or beyond the patch)
#pragma unroll
    for (int s = 0; s < IPS; ++s) {
        const int p = s * 64 + lane;
        const int iy = y0 - 1 + p / PWH, ix = x0 - 1 + p % PWH;
        poff[s] = (p < NPIX && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W) ? ((long long)bb * a.H + iy) * a.W + ix : -1;
    }
    // instruction k (0 .. XI-1) of this wave for chunk c: global instruction i = wave * XI + k -> slot q = 2 wave + k / IPS,
    // segment k % IPS (XI = 2 IPS: the segment is a compile-time constant, so poff[] stays in registers -- a runtime index
    // would put it in scratch, and a scratch load is a VMEM operation that would break the counted vmcnt waits below)
    auto issue_x = [&](int c, auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int seg = k % IPS;
        const int q = 2 * wave + k / IPS;
        const int ch = c * 64 + q * 8;
        const long long po = poff[seg];
        const h16 *src = a.zero;
        if (po >= 0 && ch < Ctot) src = (ch < a.C1) ? a.x1 + po * a.ldx1 + ch : a.x2 + po * a.ldx2 + (ch - a.C1);
        __builtin_amdgcn_global_load_lds(L2D_GPTR(src), L2D_LPTR(pb0 + (c & 1) * PBUF + (q * NPIXP + seg * 64) * 8), 16, 0, 0);
    };

    // ---- prologue: patch chunk 0, weight stages 0 and 1
    issue_x(0, std::integral_constant<int, 0>{});
    if constexpr (XI > 1) issue_x(0, std::integral_constant<int, 1>{});
    if constexpr (XI > 2) issue_x(0, std::integral_constant<int, 2>{});
    if constexpr (XI > 3) issue_x(0, std::integral_constant<int, 3>{});
    if constexpr (XI > 4) issue_x(0, std::integral_constant<int, 4>{});
    if constexpr (XI > 5) issue_x(0, std::integral_constant<int, 5>{});
    issue_w();
    issue_w();

    // ---- fragment offsets (halfs): weights as igemm (row-major 64 x 64 tile, source-side XOR swizzle); patch: 16-byte slot
    // q = kk * 4 + lg of pixel (py + dy, px + dx), stored at (q * NPIXP + pixel) * 16 bytes
    int aoff[2][2], boff[2][MI];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wn * 32 + i * 16 + li;
            aoff[kk][i] = r * BK + (((kk * 4 + lg) ^ (r & 7)) << 3);
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int tt = (wm * MI + j) * 16 + li;
            boff[kk][j] = ((kk * 4 + lg) * NPIXP + (tt / PW) * PWH + (tt % PW)) * 8;
        }
    }
    f32x4 acc[2][MI];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs_slot = 0, st_idx = 0;
    for (int c = 0; c < NCH; ++c) {
        const bool more_x = c + 1 < NCH;
        const h16 *pb = pb0 + (c & 1) * PBUF;
        auto stage = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            // this stage's weights were requested two stages ago; younger loads = the next weight stage (2, unless this is the
            // last stage) + the patch instructions of the last two stages (one each for taps < XI of a chunk that has a successor)
            constexpr int XA = ((t >= 1 && t - 1 < XI) ? 1 : 0) + ((t >= 2 && t - 2 < XI) ? 1 : 0);
            if (more_x) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + XA) : "memory");
            else if (st_idx + 1 < total) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // every wave's share of this stage has landed; everyone finished the previous stage
            if (st_idx + 2 < total) issue_w();
            if constexpr (t < XI) { if (more_x) issue_x(c + 1, std::integral_constant<int, t>{}); }
            const h16 *ws = ring + cs_slot * WST;
            constexpr int tapoff = ((t / 3) * PWH + (t % 3)) * 8;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                h16x8 af[2], bf[MI];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = l2d_ld8(ws + aoff[kk][i]);
#pragma unroll
                for (int j = 0; j < MI; ++j) bf[j] = l2d_ld8(pb + boff[kk][j] + tapoff);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            cs_slot = (cs_slot + 1 == NSW) ? 0 : cs_slot + 1;
            ++st_idx;
        };
        stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{}); stage(std::integral_constant<int, 8>{});
    }

    // ---------------------------------------------------------------- epilogue (through LDS, as igemm.hip)
    constexpr int pitch = TN + 8;
    h16 *ot = smem;
    // (time-embedding bias: one row per sample in the stream, one row for all frames of a warm-up pass)
    const float *rb = a.rowbias ? a.rowbias + (long long)((bb * a.H * a.W) / a.rows_per_bias) * a.ldrb : nullptr;
    __syncthreads();                                   // every wave is done with the patch and the ring
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = wn * 32 + i * 16 + lg * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bv = *reinterpret_cast<const f32x4 *>(a.bias + n0 + col);
        if (rb) bv += *reinterpret_cast<const f32x4 *>(rb + n0 + col);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int row = (wm * MI + j) * 16 + li;
            f32x4 v = acc[i][j] + bv;
            if (a.epi == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (a.epi == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = l2d_silu(v[r]);
            } else if (a.epi == 5) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = l2d_gelu(v[r]);
            }
            h16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (h16)v[r];
            *reinterpret_cast<h16x4 *>(ot + row * pitch + col) = o;
        }
    }
    constexpr int CPRN = TN / 8, EPI_IT = (TM * CPRN) / 256;       // 8 chunks of 16 bytes per row
    h16x8 resv[EPI_IT > 0 ? EPI_IT : 1];
    auto mrow = [&](int row) { return ((long long)bb * a.H + y0 + row / PW) * a.W + x0 + row % PW; };
    if (a.res) {
#pragma unroll
        for (int it = 0; it < EPI_IT; ++it) {
            const int cidx = it * 256 + tid, row = cidx / CPRN, cc = cidx % CPRN;
            resv[it] = l2d_ld8(a.res + mrow(row) * a.ldr + n0 + cc * 8);
        }
    }
    __syncthreads();
    const bool gn = a.gn1 != nullptr;
    float gs[4], gq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { gs[e] = 0.f; gq[e] = 0.f; }
    const h16x2 ones2 = {(h16)1.0f, (h16)1.0f};
#pragma unroll
    for (int it = 0; it < EPI_IT; ++it) {
        const int cidx = it * 256 + tid, row = cidx / CPRN, cc = cidx % CPRN;
        h16x8 v = l2d_ld8(ot + row * pitch + cc * 8);
        if (a.res) v = v + resv[it];
        if (a.epi == 4) v = __builtin_elementwise_max(v, l2d_zero8());          // relu(conv + skip), TAESD blocks
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
    if (gn) {
        __syncthreads();                                            // every thread is done reading the staged tile
        float *red = reinterpret_cast<float *>(smem);               // [256][8]: 4 pair sums | 4 pair sums of squares
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = gs[e]; red[tid * 8 + 4 + e] = gq[e]; }
        __syncthreads();
        float *chs1 = red + 256 * 8, *chs2 = chs1 + TN / 2;
        if (tid < TN / 2) {
            const int cc = tid >> 2, e = tid & 3;
            float s = 0.f, q = 0.f;
            for (int r = 0; r < 256 / CPRN; ++r) { s += red[(r * CPRN + cc) * 8 + e]; q += red[(r * CPRN + cc) * 8 + 4 + e]; }
            chs1[tid] = s; chs2[tid] = q;
        }
        __syncthreads();
        l2d_gn_flush(a.gn1, a.gnG, a.cpg1 >> 1, a.choff1 >> 1, bb, chs1, chs2, n0 >> 1, TN >> 1, tid);
        l2d_gn_flush(a.gn2, a.gnG, a.cpg2 >> 1, a.choff2 >> 1, bb, chs1, chs2, n0 >> 1, TN >> 1, tid);
    }
}

template <int PH, int PW>
static void launch_pc(const PConvArgs &a, hipStream_t s) {
    constexpr int NPIXP = ((PH + 2) * (PW + 2) + 63) / 64 * 64;
    constexpr size_t RING = (size_t)2 * 8 * NPIXP * 16 + (size_t)3 * 64 * 64 * 2;
    constexpr size_t EPI = (size_t)(256 * 8 + 64) * 4;             // GroupNorm reduction scratch (the staged tile is smaller than the ring)
    constexpr size_t LDS = RING > EPI ? RING : EPI;
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (LDS > 65536 && !attr_done) {
        if (hipFuncSetAttribute((const void *)pconv_kernel<PH, PW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) == hipSuccess) attr_done = true;
        else (void)hipGetLastError();
    }
    hipLaunchKernelGGL((pconv_kernel<PH, PW>), dim3(a.nwg), dim3(256), LDS, s, a);
}

int l2d_launch_pconv(const l2d_op *op, hipStream_t s) {
    PConvArgs a;
    a.x1 = (const h16 *)op->p[0]; a.x2 = (const h16 *)op->p[1]; a.w = (const h16 *)op->p[2];
    a.bias = (const float *)op->p[3]; a.rowbias = (const float *)op->p[4]; a.res = (const h16 *)op->p[5];
    a.out = (h16 *)op->p[6]; a.zero = (const h16 *)op->p[7];
    a.gn1 = (unsigned long long *)op->p[9]; a.gn2 = (unsigned long long *)op->p[10];
    a.C1 = op->i[1]; a.C2 = op->i[2]; a.ldx1 = op->i[3]; a.ldx2 = op->i[4]; a.CinP = op->i[5]; a.B = op->i[6];
    a.H = op->i[7]; a.W = op->i[8]; a.Nout = op->i[14]; a.ldo = op->i[15]; a.ldr = op->i[16]; a.ldrb = op->i[17];
    a.rows_per_bias = op->i[18]; a.epi = op->i[19];
    const int PH = op->i[9], PW = op->i[10];
    a.order = op->i[11] ? 1 : 0;
    a.gnT = op->i[24]; a.gnG = op->i[25]; a.cpg1 = op->i[26]; a.choff1 = op->i[27]; a.cpg2 = op->i[28]; a.choff2 = op->i[29];
    if (!a.gn1 && a.gn2) { a.gn1 = a.gn2; a.cpg1 = a.cpg2; a.choff1 = a.choff2; a.gn2 = nullptr; }
    const bool pok = (PH == 8 && PW == 16) || (PH == 8 && PW == 8) || (PH == 4 && PW == 8);
    if (!a.x1 || !a.w || !a.out || !a.zero || !pok || a.B <= 0 || a.H <= 0 || a.W <= 0 || (a.H % PH) || (a.W % PW) ||
        a.C1 <= 0 || (a.C1 % 64) || (a.C2 % 64) || (a.C2 > 0 && !a.x2) || a.CinP != a.C1 + a.C2 || (a.Nout % 64) || a.Nout <= 0 ||
        (a.ldx1 % 8) || a.ldx1 < a.C1 || (a.C2 > 0 && ((a.ldx2 % 8) || a.ldx2 < a.C2)) || (a.ldo % 8) || (a.res && (a.ldr % 8)) ||
        (a.rowbias && (a.ldrb <= 0 || a.rows_per_bias <= 0)) || a.epi < 0 || a.epi > 5 || a.epi == 1 || (a.epi == 4 && !a.res) ||
        (((unsigned long long)a.x1 | (unsigned long long)a.x2 | (unsigned long long)a.w | (unsigned long long)a.out |
          (unsigned long long)a.res | (unsigned long long)a.bias | (unsigned long long)a.rowbias) & 15)) {
        l2d_set_error("pconv(tag %d): invalid arguments (B=%d H=%d W=%d C1=%d C2=%d CinP=%d Nout=%d patch %dx%d)", op->tag, a.B, a.H,
                      a.W, a.C1, a.C2, a.CinP, a.Nout, PH, PW);
        return L2D_EINVAL;
    }
    if (a.gn1) {
        if (a.gnT != a.H * a.W || a.gnG <= 0 || a.gnG > 32 || a.cpg1 <= 0 || (a.gn2 && a.cpg2 <= 0) || ((a.cpg1 | a.choff1) & 1) ||
            (a.gn2 && ((a.cpg2 | a.choff2) & 1))) {
            l2d_set_error("pconv(tag %d): GroupNorm statistics need T == H * W, even group sizes and offsets", op->tag);
            return L2D_EINVAL;
        }
    }
    a.npx = a.W / PW; a.npy = a.H / PH; a.ntn = a.Nout / 64;
    a.nwg = a.B * a.npx * a.npy * a.ntn;
    // (pixel indices are 32-bit, every element offset is formed in 64 bits)
    if ((long long)a.B * a.H * a.W >= (1ll << 28) || (long long)a.nwg >= (1ll << 30)) {
        l2d_set_error("pconv(tag %d): tensor too large for the kernel's index arithmetic", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    if (PH == 8 && PW == 16) launch_pc<8, 16>(a, s);
    else if (PH == 8 && PW == 8) launch_pc<8, 8>(a, s);
    else launch_pc<4, 8>(a, s);
    return l2d_check_launch("pconv", op->tag);
}
