// GroupNorm (+SiLU) and LayerNorm for channels-last fp16 activations on gfx950.  HBM-bound:
// 16-byte vector loads, fp32 statistics, wave-shuffle reductions, no intermediate layout copies.
//
// GroupNorm replaces nn.GroupNorm / InflatedGroupNorm (+ F.silu) of the reference (resnet.py:68-76,
// 232-233,243,249; attention.py:57,102; motion_module.py:181,273; unet_depth_streaming.py:620-621).
// It reads up to two inputs as one virtual channel-concat [x1 | x2], so the skip-connection torch.cat
// of the up blocks (unet_blocks_streaming.py:683,814) is never materialised.
//   pass 1 (gn_stats): per-(b, pixel-chunk) partial sum / sum-of-squares for each of the G groups (fixed summation order)
//   pass 2 (gn_apply): deterministic reduction of the partials, then y = (x-mean)*rstd*gamma+beta (-> SiLU)
// LayerNorm replaces nn.LayerNorm (attention.py:182,199,205; motion_module.py:355,361): one wave per row.
#include <stdlib.h>

#include "common.h"

struct GNArgs {
    const h16 *x1, *x2;
    float *partial;
    const h16 *gamma, *beta;
    h16 *out;
    int B, T, C1, C2, ld1, ld2, G, nchunk, silu;
    float eps;
    const h16 *res;           // optional residual [B*T][C], added AFTER the normalisation and before the activation (act 3)
    const long long *acc;     // nchunk == 0: fixed-point statistics [B][G][2] accumulated by the producing GEMMs (igemm.hip)
};

// thread -> (pixel row within the pass, 8-channel vector column); returns false if idle
__device__ __forceinline__ h16x8 gn_load(const GNArgs &a, long long pix, int vc) {
    int c = vc * 8;
    const h16 *src = (c < a.C1) ? a.x1 + pix * a.ld1 + c : a.x2 + pix * a.ld2 + (c - a.C1);
    return l2d_ld8(src);
}

__global__ __launch_bounds__(256) void gn_stats_kernel(GNArgs a) {
    // Deterministic: per-thread channel sums go to LDS and thread g adds up group g's entries in a fixed order (no
    // float atomics anywhere), so a frame is bit-repeatable and a hipGraph replay equals the direct launches exactly.
    __shared__ float s_part[256][17];              // [thread][8 sums | 8 sums of squares] (+1: bank spread)
    const int tid = threadIdx.x;
    const int C = a.C1 + a.C2, nvc = C / 8, cpg = C / a.G;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (a.T + a.nchunk - 1) / a.nchunk;
    const int t0 = chunk * per, t1 = min(a.T, t0 + per);
    const int cols = min(nvc, 256), PR = 256 / cols;
    const int prow = tid / cols, vcl = tid - prow * cols;
    float gs = 0.f, gq = 0.f;                      // partial sums of group tid/8 (8 threads per group), over all passes
    for (int cb = 0; cb < nvc; cb += cols) {
        const int vc = cb + vcl;
        const bool active = prow < PR && vc < nvc;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        if (active) {
            // 4 rows per trip, loads first: the loop is one dependent HBM/L2 round trip per iteration otherwise
            for (int t = t0 + prow; t < t1; t += 4 * PR) {
                h16x8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tt = t + u * PR;
                    v[u] = tt < t1 ? gn_load(a, (long long)b * a.T + tt, vc) : l2d_zero8();
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { float f = (float)v[u][e]; s[e] += f; q[e] += f * f; }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s_part[tid][e] = s[e]; s_part[tid][8 + e] = q[e]; }
        __syncthreads();
        {
            // 8 threads per group (G <= 32): thread (g, j) adds the group's (channel, pixel-row) entries k = j, j+8, ...
            // of this pass: channels [max(g*cpg, cb*8), min((g+1)*cpg, (cb+cols)*8, C)) x PR pixel rows
            const int g = tid >> 3, j = tid & 7;
            if (g < a.G) {
                const int c0 = max(g * cpg, cb * 8), c1 = min(min((g + 1) * cpg, (cb + cols) * 8), C);
                const int n = max(c1 - c0, 0) * PR;
                for (int k = j; k < n; k += 8) {
                    const int c = c0 + k / PR, pr = k - (k / PR) * PR;
                    const int row = pr * cols + (c >> 3) - cb, e = c & 7;
                    gs += s_part[row][e];
                    gq += s_part[row][8 + e];
                }
            }
        }
        __syncthreads();
    }
    // fixed-order combine of the 8 partials of each group (lanes 8g .. 8g+7 of one wave)
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { gs += __shfl_xor(gs, o, 64); gq += __shfl_xor(gq, o, 64); }
    if ((tid & 7) == 0 && (tid >> 3) < a.G) {
        float *dst = a.partial + (((long long)b * a.nchunk + chunk) * a.G + (tid >> 3)) * 2;
        dst[0] = gs;
        dst[1] = gq;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(GNArgs a, int pix_per_block) {
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ float s_part[4][64][2];
    const int tid = threadIdx.x;
    const int C = a.C1 + a.C2, nvc = C / 8, cpg = C / a.G;
    const int b = blockIdx.y;
    // The block's operands (gamma, beta, four rows of x) do not depend on the statistics: they are requested HERE, so that
    // the launch is one memory round trip deep (statistics, parameters and data in flight together) instead of three
    // dependent ones -- these launches move 0.2-5 MB and sit on the latency floor.  The launcher sizes a block to exactly the
    // four pixel-row passes requested here and gives every 2048-channel band its own blocks (blockIdx.z), so no block has a
    // second, dependent trip (round 6: the old 16 KB blocks ran a second trip of one or two rows behind the first).
    const int t0 = blockIdx.x * pix_per_block, t1 = min(a.T, t0 + pix_per_block);
    const int cols = min(nvc, 256), PR = 256 / cols;
    const int prow = tid / cols, vcl = tid - prow * cols;
    const int vc = blockIdx.z * cols + vcl;
    const bool act0 = prow < PR && vc < nvc;
    h16x8 gm = l2d_zero8(), bt = l2d_zero8(), v0[4];
    if (act0) {
        gm = l2d_ld8(a.gamma + vc * 8);
        bt = l2d_ld8(a.beta + vc * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = t0 + prow + u * PR;
            v0[u] = tt < t1 ? gn_load(a, (long long)b * a.T + tt, vc) : l2d_zero8();
        }
    }
    if (a.nchunk == 0) {   // statistics arrive as integers in units of 2^-20 (sum) and 2^-12 (sum of squares)
        const int g = tid & 63, part = tid >> 6;
        float s = 0.f, q = 0.f;
        if (g < a.G && part == 0) {
            const long long *src = a.acc + ((long long)b * a.G + g) * 2;
            s = (float)((double)src[0] * (1.0 / 1048576.0));
            q = (float)((double)src[1] * (1.0 / 4096.0));
        }
        s_part[part][g][0] = s; s_part[part][g][1] = q;
    } else {   // deterministic reduction of the per-chunk partials: 4 slices x G groups in parallel, fixed order
        const int g = tid & 63, part = tid >> 6;
        float s = 0.f, q = 0.f;
        if (g < a.G) {
            const float2 *src = reinterpret_cast<const float2 *>(a.partial) + (long long)b * a.nchunk * a.G + g;
#pragma unroll 8
            for (int ch = part; ch < a.nchunk; ch += 4) {      // unrolled: the partial loads go out together
                const float2 v = src[(long long)ch * a.G];
                s += v.x; q += v.y;
            }
        }
        s_part[part][g][0] = s; s_part[part][g][1] = q;
    }
    __syncthreads();
    if (tid < a.G) {
        float s = (s_part[0][tid][0] + s_part[1][tid][0]) + (s_part[2][tid][0] + s_part[3][tid][0]);
        float q = (s_part[0][tid][1] + s_part[1][tid][1]) + (s_part[2][tid][1] + s_part[3][tid][1]);
        float inv = 1.0f / ((float)a.T * (float)cpg);
        float mean = s * inv;
        float var = fmaxf(q * inv - mean * mean, 0.f);
        s_mean[tid] = mean;
        s_rstd[tid] = rsqrtf(var + a.eps);
    }
    __syncthreads();
    if (!act0) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int g = (vc * 8 + e) / cpg;
        sc[e] = s_rstd[g] * (float)gm[e];
        sh[e] = (float)bt[e] - s_mean[g] * sc[e];
    }
    for (int t = t0 + prow; t < t1; t += 4 * PR) {     // 4 rows per trip, loads first (see gn_stats); one trip as launched
        h16x8 v[4];
        if (t == t0 + prow) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = v0[u];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + u * PR;
                v[u] = tt < t1 ? gn_load(a, (long long)b * a.T + tt, vc) : l2d_zero8();
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = t + u * PR;
            if (tt >= t1) break;
            h16x8 o, rr = l2d_zero8();
            const long long oidx = ((long long)b * a.T + tt) * C + vc * 8;
            if (a.silu == 3) rr = l2d_ld8(a.res + oidx);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y = (float)v[u][e] * sc[e] + sh[e];
                if (a.silu == 1) y = l2d_silu(y);
                else if (a.silu == 2) y = fmaxf(y, 0.f);
                else if (a.silu == 3) y = fmaxf((float)(h16)y + (float)rr[e], 0.f);   // relu(norm(x) + shortcut): ResNetV2 bottleneck
                o[e] = (h16)y;
            }
            l2d_st8(a.out + oidx, o);
        }
    }
}

// One-launch GroupNorm for SMALL tensors (round 6, third session): gn_apply with nchunk == 0 and no accumulator.  Levels whose tokens
// per sample are no whole number of GEMM tiles (12 x 12, 6 x 6, 10 x 10 ... pixels: every resolution outside the tuned ones) cannot take
// their statistics from the producers' epilogues and paid two launches per GroupNorm (gn_stats + gn_apply: 2 x ~4.8 us for 20-370 KB).
// Here a block owns (sample, band of whole groups = lcm(cpg, 8) channels) over ALL T pixel rows: its T * band / 8 vectors (at most 16
// per thread) stay in registers between the statistics and the apply pass -- one read, one launch.  Thread -> (fixed 8-channel column,
// pixel rows prow + j * PR), so the group of each of its 8 channels is a per-thread constant; per-thread sums meet through a fixed
// shuffle tree and four LDS slots (no float atomics: bit-repeatable).  Same statistics formula and apply arithmetic as gn_apply.
__global__ __launch_bounds__(256) void gn_self_kernel(GNArgs a, int band) {
    constexpr int NV = 16, NGB = 4;                 // vectors per thread, groups per band (launcher checks both bounds)
    __shared__ float s_w[4][2 * NGB];
    __shared__ float s_mean[NGB], s_rstd[NGB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C1 + a.C2, cpg = C / a.G;
    const int b = blockIdx.y, nvb = band >> 3, PR = 256 / nvb, ng = band / cpg;
    const int prow = tid / nvb, vcl = tid - prow * nvb;
    const bool act = prow < PR;
    const int vc = blockIdx.x * nvb + vcl;
    int ge[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ge[e] = (vcl * 8 + e) / cpg;
    h16x8 gm = l2d_zero8(), bt = l2d_zero8(), v[NV];
    if (act) {
        gm = l2d_ld8(a.gamma + vc * 8);
        bt = l2d_ld8(a.beta + vc * 8);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {                  // every load of the block goes out before the first sum
        const int t = prow + j * PR;
        v[j] = (act && t < a.T) ? gn_load(a, (long long)b * a.T + t, vc) : l2d_zero8();
    }
    float se[8], qe[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { se[e] = 0.f; qe[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[j][e]; se[e] += f; qe[e] = fmaf(f, f, qe[e]); }
    float sg[NGB], qg[NGB];
#pragma unroll
    for (int g = 0; g < NGB; ++g) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s += ge[e] == g ? se[e] : 0.f; q += ge[e] == g ? qe[e] : 0.f; }
        sg[g] = l2d_wave_sum(s);
        qg[g] = l2d_wave_sum(q);
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < NGB; ++g) { s_w[wave][g] = sg[g]; s_w[wave][NGB + g] = qg[g]; }
    }
    __syncthreads();
    if (tid < ng) {
        const float s = (s_w[0][tid] + s_w[1][tid]) + (s_w[2][tid] + s_w[3][tid]);
        const float q = (s_w[0][NGB + tid] + s_w[1][NGB + tid]) + (s_w[2][NGB + tid] + s_w[3][NGB + tid]);
        const float inv = 1.0f / ((float)a.T * (float)cpg);
        const float mean = s * inv;
        const float var = fmaxf(q * inv - mean * mean, 0.f);
        s_mean[tid] = mean;
        s_rstd[tid] = rsqrtf(var + a.eps);
    }
    __syncthreads();
    if (!act) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = s_rstd[ge[e]] * (float)gm[e];
        sh[e] = (float)bt[e] - s_mean[ge[e]] * sc[e];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int t = prow + j * PR;
        if (t >= a.T) break;
        h16x8 o, rr = l2d_zero8();
        const long long oidx = ((long long)b * a.T + t) * C + vc * 8;
        if (a.silu == 3) rr = l2d_ld8(a.res + oidx);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)v[j][e] * sc[e] + sh[e];
            if (a.silu == 1) y = l2d_silu(y);
            else if (a.silu == 2) y = fmaxf(y, 0.f);
            else if (a.silu == 3) y = fmaxf((float)(h16)y + (float)rr[e], 0.f);
            o[e] = (h16)y;
        }
        l2d_st8(a.out + oidx, o);
    }
}

// band (channels) of the one-launch form for this shape, or 0 when it does not fit: whole groups, a multiple of 8 channels, at most 4
// groups and 128 channels per band, at most 16 vectors per thread
static int gn_self_band(int T, int C, int G) {
    if (G <= 0 || C % G) return 0;
    const int cpg = C / G;
    int band = cpg;
    while (band % 8) band += cpg;                   // lcm(cpg, 8)
    if (band > 128 || band / cpg > 4 || C % band) return 0;
    const int pr = 256 / (band / 8);
    return (T + pr - 1) / pr <= 16 ? band : 0;
}

static int gn_args(const l2d_op *op, GNArgs &a, bool apply) {
    a.x1 = (const h16 *)op->p[0]; a.x2 = (const h16 *)op->p[1]; a.partial = (float *)op->p[2];
    a.gamma = (const h16 *)op->p[3]; a.beta = (const h16 *)op->p[4]; a.out = (h16 *)op->p[5];
    a.B = op->i[0]; a.T = op->i[1]; a.C1 = op->i[2]; a.C2 = op->i[3]; a.ld1 = op->i[4]; a.ld2 = op->i[5];
    a.G = op->i[6]; a.nchunk = op->i[7]; a.silu = op->i[8]; a.eps = op->f[0];
    a.acc = (const long long *)op->p[6];
    a.res = (const h16 *)op->p[7];
    int C = a.C1 + a.C2;
    if (apply && a.nchunk == 0 && a.acc) a.partial = (float *)a.acc;      // accumulator mode: no partial buffer
    const bool self = apply && a.nchunk == 0 && !a.acc;                   // one-launch form: statistics inside the launch
    if (self) a.partial = (float *)a.x1;                                  // (unused; passes the pointer check below)
    if (self && !gn_self_band(a.T, C, a.G)) {
        l2d_set_error("groupnorm(tag %d): nchunk = 0 without accumulators is the one-launch form for small tensors (whole groups in bands of "
                      "<= 128 channels, <= 4 groups per band, T * band / 8 <= 4096 vectors); T=%d C=%d G=%d does not fit", op->tag, a.T, C, a.G);
        return L2D_EINVAL;
    }
    if (!a.x1 || !a.partial || a.B <= 0 || a.T <= 0 || a.G <= 0 || a.G > 32 || (C % a.G) || (a.C1 % 8) || (a.C2 % 8) ||
        (a.C2 > 0 && !a.x2) || a.nchunk < 0 || (a.nchunk == 0 && !apply) || a.nchunk > a.T || C / 8 > 512 || (a.ld1 % 8) || (a.C2 > 0 && (a.ld2 % 8)) ||
        (apply && (!a.gamma || !a.beta || !a.out || a.silu < 0 || a.silu > 3 || (a.silu == 3 && !a.res)))) {
        l2d_set_error("groupnorm(tag %d): invalid arguments (B=%d T=%d C1=%d C2=%d G=%d nchunk=%d)", op->tag, a.B, a.T,
                      a.C1, a.C2, a.G, a.nchunk);
        return L2D_EINVAL;
    }
    return L2D_OK;
}

int l2d_launch_gn_stats(const l2d_op *op, hipStream_t s) {
    GNArgs a;
    int rc = gn_args(op, a, false);
    if (rc) return rc;
    L2D_DRY_RETURN();
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a.nchunk, a.B), dim3(256), 0, s, a);
    return l2d_check_launch("gn_stats", op->tag);
}

int l2d_launch_gn_apply(const l2d_op *op, hipStream_t s) {
    GNArgs a;
    int rc = gn_args(op, a, true);
    if (rc) return rc;
    L2D_DRY_RETURN();
    int C = a.C1 + a.C2;
    if (a.nchunk == 0 && !a.acc) {
        const int band = gn_self_band(a.T, C, a.G);
        hipLaunchKernelGGL(gn_self_kernel, dim3(C / band, a.B), dim3(256), 0, s, a, band);
        return l2d_check_launch("gn_self", op->tag);
    }
    // One trip per block: the four passes of pixel rows the kernel requests before it looks at the statistics (8-32 KB of
    // activations per block; 340-1000 blocks at the 64 x 64 level), one grid plane per band of 2048 channels.  L2D_GN_BLOCK16K=1
    // restores the ~16 KB blocks of rounds 3-5 (a second, dependent trip of a row or two) for A/B: profiles/round6_s_*.
    static int old_blocks = -1;
    if (old_blocks < 0) { const char *e = getenv("L2D_GN_BLOCK16K"); old_blocks = e ? atoi(e) : 0; }
    int cols = (C / 8) < 256 ? (C / 8) : 256;
    int pr = 256 / cols;
    int ppb = old_blocks ? (8192 + C - 1) / C : 4 * pr;
    if (ppb < pr) ppb = pr;
    if (ppb > a.T) ppb = a.T;
    int nb = (a.T + ppb - 1) / ppb;
    int nz = (C / 8 + cols - 1) / cols;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nb, a.B, nz), dim3(256), 0, s, a, ppb);
    return l2d_check_launch("gn_apply", op->tag);
}

// ------------------------------------------------------------------------------------------- LayerNorm
__global__ __launch_bounds__(256) void layernorm_kernel(const h16 *__restrict__ x, const h16 *__restrict__ gamma,
                                                        const h16 *__restrict__ beta, h16 *__restrict__ out, int rows,
                                                        int C, int ldx, int ldo, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvc = C / 8;
    constexpr int MAXV = 4;  // C <= 2048
    h16x8 v[MAXV], gmv[MAXV], btv[MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {      // all loads of the row AND its affine parameters go out before the first reduction
        int vc = lane + 64 * j;
        if (vc < nvc) { gmv[j] = l2d_ld8(gamma + vc * 8); btv[j] = l2d_ld8(beta + vc * 8); }
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        int vc = lane + 64 * j;
        if (vc < nvc) {
            v[j] = l2d_ld8(x + (long long)row * ldx + vc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[j][e];
        }
    }
    float mean = l2d_wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        int vc = lane + 64 * j;
        if (vc < nvc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float d = (float)v[j][e] - mean; q += d * d; }
        }
    }
    float rstd = rsqrtf(l2d_wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        int vc = lane + 64 * j;
        if (vc < nvc) {
            const h16x8 gm = gmv[j], bt = btv[j];
            h16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)(((float)v[j][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
            l2d_st8(out + (long long)row * ldo + vc * 8, o);
        }
    }
}

int l2d_launch_layernorm(const l2d_op *op, hipStream_t s) {
    const h16 *x = (const h16 *)op->p[0], *g = (const h16 *)op->p[1], *b = (const h16 *)op->p[2];
    h16 *out = (h16 *)op->p[3];
    int rows = op->i[0], C = op->i[1], ldx = op->i[2], ldo = op->i[3];
    if (!x || !g || !b || !out || rows <= 0 || C <= 0 || (C % 8) || C > 2048 || (ldx % 8) || (ldo % 8)) {
        l2d_set_error("layernorm(tag %d): invalid arguments (rows=%d C=%d)", op->tag, rows, C);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, out, rows, C, ldx, ldo, op->f[0]);
    return l2d_check_launch("layernorm", op->tag);
}
