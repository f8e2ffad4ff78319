// Implicit-GEMM on MFMA for gfx950: nn.Linear / 1x1 conv / 3x3 conv (stride 1|2, nearest-2x upsample and
// channel-concat folded into the gather) with fused epilogue (bias, per-sample time-embedding bias,
// residual, GEGLU).  Replaces the cuDNN/cuBLAS calls behind the reference's InflatedConv3d / nn.Linear
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
//   * 256 threads = 4 waves (2x2), block tile TN x TM (128x128 or 64x64), BK = 64.
//   * LDS: [rows][64] halfs (128 B rows), 16-byte slots XOR-swizzled with (row & 7): ds_read_b128
//     fragment reads and ds_write_b128 staging writes are both bank-conflict free (simulated against
//     the gfx950 lane-group table); two buffers, register-staged prefetch of tile k+1 while tile k is
//     on the matrix cores; one barrier per K step.
#include "common.h"

#define BK 64

struct IGemmArgs {
    const h16 *x1, *x2, *w;
    const float *bias, *rowbias;
    const h16 *res;
    h16 *out;
    int taps, C1, C2, ldx1, ldx2, CinP, B, Hin, Win, Hout, Wout, stride, ups;
    int M, Nout, ldo, ldr, ldrb, rows_per_bias, epi, Kp;
    long long sx1, sw, so, sres;
};

template <int TN, int TM, int TAPS>
__global__ __launch_bounds__(256) void igemm_kernel(IGemmArgs a) {
    constexpr int NI = TN / 32;  // 16-row fragments per wave along channels
    constexpr int MI = TM / 32;  // 16-col fragments per wave along tokens
    constexpr int WCH = TN / 32; // staging chunks per thread (weights)
    constexpr int XCH = TM / 32; // staging chunks per thread (tokens)
    __shared__ __attribute__((aligned(16))) h16 Ws[2][TN * BK];
    __shared__ __attribute__((aligned(16))) h16 Xs[2][TM * BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    const int ntn = (a.Nout + TN - 1) / TN;
    const int tile_n = blockIdx.x % ntn;
    const int tile_m = blockIdx.x / ntn;
    const int n0 = tile_n * TN, m0 = tile_m * TM;
    const long long z = blockIdx.z;
    const h16 *x1 = a.x1 + z * a.sx1;
    const h16 *wp = a.w + z * a.sw;
    h16 *outp = a.out + z * a.so;
    const h16 *resp = a.res ? a.res + z * a.sres : nullptr;

    const int slot = tid & 7;
    const int row0 = tid >> 3;  // + 32*j
    const int Ctot = a.C1 + a.C2;

    // per-thread token-row descriptors for the gather
    int xb[XCH], xy[XCH], xx[XCH];
    bool xv[XCH];
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
        int m = m0 + row0 + 32 * j;
        xv[j] = m < a.M;
        if (TAPS == 9) {
            int hw = a.Hout * a.Wout;
            int b = m / hw, r = m - b * hw;
            xb[j] = b;
            xy[j] = r / a.Wout;
            xx[j] = r - xy[j] * a.Wout;
        } else {
            xb[j] = m; xy[j] = 0; xx[j] = 0;
        }
    }
    bool wv[WCH];
#pragma unroll
    for (int j = 0; j < WCH; ++j) wv[j] = (n0 + row0 + 32 * j) < a.Nout;

    h16x8 wreg[WCH], xreg[XCH];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            int n = n0 + row0 + 32 * j;
            wreg[j] = wv[j] ? l2d_ld8(wp + (long long)n * a.Kp + k0 + slot * 8) : l2d_zero8();
        }
        int c, ky = 0, kx = 0;
        if (TAPS == 9) {
            int tap = k0 / a.CinP;
            c = k0 - tap * a.CinP + slot * 8;
            ky = tap / 3; kx = tap - ky * 3;
        } else {
            c = k0 + slot * 8;
        }
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            bool ok = xv[j] && c < Ctot;
            long long pix;
            if (TAPS == 9) {
                int iy = xy[j] * a.stride + ky - 1;
                int ix = xx[j] * a.stride + kx - 1;
                ok = ok && iy >= 0 && ix >= 0 && iy < (a.Hin << a.ups) && ix < (a.Win << a.ups);
                pix = ((long long)xb[j] * a.Hin + (iy >> a.ups)) * a.Win + (ix >> a.ups);
            } else {
                pix = xb[j];
            }
            h16x8 v = l2d_zero8();
            if (ok) {
                const h16 *src = (c < a.C1) ? x1 + pix * a.ldx1 + c : a.x2 + pix * a.ldx2 + (c - a.C1);
                v = l2d_ld8(src);
            }
            xreg[j] = v;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            int r = row0 + 32 * j;
            l2d_st8(&Ws[buf][r * BK + ((slot ^ (r & 7)) << 3)], wreg[j]);
        }
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            int r = row0 + 32 * j;
            l2d_st8(&Xs[buf][r * BK + ((slot ^ (r & 7)) << 3)], xreg[j]);
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = a.Kp / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h16x8 af[NI], bf[MI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int r = wn * (TN / 2) + i * 16 + li;
                af[i] = l2d_ld8(&Ws[cur][r * BK + (((kk * 4 + lg) ^ (r & 7)) << 3)]);
            }
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                int r = wm * (TM / 2) + j * 16 + li;
                bf[j] = l2d_ld8(&Xs[cur][r * BK + (((kk * 4 + lg) ^ (r & 7)) << 3)]);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---------------------------------------------------------------- epilogue
    if (a.epi == 1) {
        // GEGLU: fragment pairs (2p, 2p+1) hold value / gate of the same output columns
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            int m = m0 + wm * (TM / 2) + j * 16 + li;
            if (m >= a.M) continue;
#pragma unroll
            for (int p = 0; p < NI / 2; ++p) {
                int nv = n0 + wn * (TN / 2) + (2 * p) * 16 + lg * 4;      // packed row of the value part
                int ng = nv + 16;                                         // packed row of the gate part
                if (ng >= a.Nout) continue;
                int no = (n0 + wn * (TN / 2)) / 2 + p * 16 + lg * 4;      // output column
                f32x4 bv = *reinterpret_cast<const f32x4 *>(a.bias + nv);
                f32x4 bg = *reinterpret_cast<const f32x4 *>(a.bias + ng);
                h16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[2 * p][j][r] + bv[r];
                    float g = acc[2 * p + 1][j][r] + bg[r];
                    o[r] = (h16)(v * l2d_gelu(g));
                }
                *reinterpret_cast<h16x4 *>(outp + (long long)m * a.ldo + no) = o;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        int m = m0 + wm * (TM / 2) + j * 16 + li;
        if (m >= a.M) continue;
        const float *rb = a.rowbias ? a.rowbias + (long long)(m / a.rows_per_bias) * a.ldrb : nullptr;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int n = n0 + wn * (TN / 2) + i * 16 + lg * 4;
            if (n >= a.Nout) continue;
            f32x4 v = acc[i][j];
            if (n + 4 > a.Nout) {
                // ragged last channel group (Nout % 4 != 0, e.g. the swapped V^T GEMM with an odd token count)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= a.Nout) break;
                    float y = v[r];
                    if (a.bias) y += a.bias[n + r];
                    if (rb) y += rb[n + r];
                    if (a.epi == 2) y = l2d_silu(y);
                    if (resp) y += (float)resp[(long long)m * a.ldr + n + r];
                    outp[(long long)m * a.ldo + n + r] = (h16)y;
                }
                continue;
            }
            if (a.bias) v += *reinterpret_cast<const f32x4 *>(a.bias + n);
            if (rb) v += *reinterpret_cast<const f32x4 *>(rb + n);
            if (a.epi == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = l2d_silu(v[r]);
            }
            if (resp) {
                h16x4 rr = *reinterpret_cast<const h16x4 *>(resp + (long long)m * a.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
            }
            h16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (h16)v[r];
            *reinterpret_cast<h16x4 *>(outp + (long long)m * a.ldo + n) = o;
        }
    }
}

template <int TN, int TM>
static int launch_t(const IGemmArgs &a, int batch, hipStream_t s) {
    int ntn = (a.Nout + TN - 1) / TN, ntm = (a.M + TM - 1) / TM;
    dim3 grid(ntn * ntm, 1, batch), block(256);
    if (a.taps == 9)
        hipLaunchKernelGGL((igemm_kernel<TN, TM, 9>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((igemm_kernel<TN, TM, 1>), grid, block, 0, s, a);
    return L2D_OK;
}

int l2d_launch_igemm(const l2d_op *op, hipStream_t s) {
    IGemmArgs a;
    a.x1 = (const h16 *)op->p[0]; a.x2 = (const h16 *)op->p[1]; a.w = (const h16 *)op->p[2];
    a.bias = (const float *)op->p[3]; a.rowbias = (const float *)op->p[4];
    a.res = (const h16 *)op->p[5]; a.out = (h16 *)op->p[6];
    a.taps = op->i[0]; a.C1 = op->i[1]; a.C2 = op->i[2]; a.ldx1 = op->i[3]; a.ldx2 = op->i[4];
    a.CinP = op->i[5]; a.B = op->i[6]; a.Hin = op->i[7]; a.Win = op->i[8]; a.Hout = op->i[9];
    a.Wout = op->i[10]; a.stride = op->i[11]; a.ups = op->i[12]; a.M = op->i[13]; a.Nout = op->i[14];
    a.ldo = op->i[15]; a.ldr = op->i[16]; a.ldrb = op->i[17]; a.rows_per_bias = op->i[18]; a.epi = op->i[19];
    int batch = op->i[20] > 0 ? op->i[20] : 1;
    a.sx1 = op->l[0]; a.sw = op->l[1]; a.so = op->l[2]; a.sres = op->l[3];
    a.Kp = a.taps * a.CinP;
    if (!a.x1 || !a.w || !a.out || (a.taps != 1 && a.taps != 9) || a.CinP <= 0 || a.CinP % BK != 0 ||
        a.M <= 0 || a.Nout <= 0 || (a.C1 % 8) || (a.C2 % 8) || (a.C2 > 0 && !a.x2) || a.C1 + a.C2 > a.CinP ||
        (a.ldo % 4) || (a.ldx1 % 8) || (a.C2 > 0 && (a.ldx2 % 8)) ||
        (a.res && (a.ldr % 4)) || (a.rowbias && a.rows_per_bias <= 0) ||
        (a.epi == 1 && (!a.bias || (a.Nout % 32))) || (a.stride != 1 && a.stride != 2) || (a.ups != 0 && a.ups != 1)) {
        l2d_set_error("igemm(tag %d): invalid arguments (taps=%d C1=%d C2=%d CinP=%d M=%d Nout=%d ldo=%d)", op->tag,
                      a.taps, a.C1, a.C2, a.CinP, a.M, a.Nout, a.ldo);
        return L2D_EINVAL;
    }
    if (a.taps == 9 && (a.M != a.B * a.Hout * a.Wout)) {
        l2d_set_error("igemm(tag %d): M != B*Hout*Wout", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    // tile choice: the big tile only when it still yields >= ~1 wave of blocks over 256 CUs
    long long big = (long long)((a.Nout + 127) / 128) * ((a.M + 127) / 128) * batch;
    int rc = (big >= 192) ? launch_t<128, 128>(a, batch, s) : launch_t<64, 64>(a, batch, s);
    if (rc != L2D_OK) return rc;
    return l2d_check_launch("igemm", op->tag);
}
