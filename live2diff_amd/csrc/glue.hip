// Depth-path glue of the per-frame pipeline (the part around the depth detector): the
// reference's `encode_depth` (pipeline_stream_animation_depth.py:544-571) does, in torch ops on fp16 tensors,
//     images_input = F.interpolate(image, (384, 384), mode="bilinear", align_corners=False)        (:553)
//     depth_map    = depth_detector(images_input)                                                   [B,384,384]
//     dn = (depth_map - depth_map.min()) / (depth_map.max() - depth_map.min())                      (:560)
//     dn = dn[:, None].repeat(1, 3, 1, 1) * 2 - 1                                                    (:561-565)
//     dn = F.interpolate(dn, (h, w), mode="bilinear", align_corners=False)                           (:566-567)
// i.e. one resize, two full-tensor reductions with host-visible results, four elementwise passes and another resize
// (9 launches and 6 intermediate tensors).  Here: `resize_bilinear` for the first line; `minmax` (two tiny launches, result
// stays on the device) + `depth_norm_resize` (normalise at the 4 taps with the reference's fp16 rounding points, blend in
// fp32, write the 3 identical channels) for the rest.  All HBM-bound elementwise work: 16-byte stores, no LDS, no MFMA.
#include "common.h"

// PyTorch's bilinear source index (align_corners=False): src = max((dst + 0.5) * in/out - 0.5, 0)
__device__ __forceinline__ void l2d_bilerp_coord(int dst, int in, int out, int &i0, int &i1, float &lam) {
    float src = ((float)dst + 0.5f) * ((float)in / (float)out) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    lam = src - (float)i0;
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const h16 *__restrict__ in, h16 *__restrict__ out, int planes, int Hin,
                                                              int Win, int Hout, int Wout) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)planes * Hout * Wout;
    if (idx >= total) return;
    const int x = (int)(idx % Wout);
    const long long r = idx / Wout;
    const int y = (int)(r % Hout);
    const long long pl = r / Hout;
    int y0, y1, x0, x1;
    float ly, lx;
    l2d_bilerp_coord(y, Hin, Hout, y0, y1, ly);
    l2d_bilerp_coord(x, Win, Wout, x0, x1, lx);
    const h16 *p = in + pl * Hin * Win;
    const float v00 = (float)p[(long long)y0 * Win + x0], v01 = (float)p[(long long)y0 * Win + x1];
    const float v10 = (float)p[(long long)y1 * Win + x0], v11 = (float)p[(long long)y1 * Win + x1];
    out[idx] = (h16)((1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11));
}

int l2d_launch_resize_bilinear(const l2d_op *op, hipStream_t s) {
    const int planes = op->i[0], Hin = op->i[1], Win = op->i[2], Hout = op->i[3], Wout = op->i[4];
    if (!op->p[0] || !op->p[1] || planes <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) {
        l2d_set_error("resize_bilinear(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    const long long total = (long long)planes * Hout * Wout;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16 *)op->p[0],
                       (h16 *)op->p[1], planes, Hin, Win, Hout, Wout);
    return l2d_check_launch("resize_bilinear", op->tag);
}

// ---- min / max of a fp16 tensor, result {min, max} as two floats on the device (no host sync)
__global__ __launch_bounds__(256) void minmax_partial_kernel(const h16 *__restrict__ x, long long n, float *__restrict__ partial) {
    float mn = 3.0e38f, mx = -3.0e38f;
    const long long n8 = n / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const h16x8 v = l2d_ld8(x + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { mn = fminf(mn, (float)v[e]); mx = fmaxf(mx, (float)v[e]); }
    }
    if (blockIdx.x == 0)
        for (long long i = n8 * 8 + threadIdx.x; i < n; i += 256) { mn = fminf(mn, (float)x[i]); mx = fmaxf(mx, (float)x[i]); }
    mn = -l2d_wave_max(-mn);
    mx = l2d_wave_max(mx);
    __shared__ float sm[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = mn; sm[4 + w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = fminf(fminf(sm[0], sm[1]), fminf(sm[2], sm[3]));
        partial[2 * blockIdx.x + 1] = fmaxf(fmaxf(sm[4], sm[5]), fmaxf(sm[6], sm[7]));
    }
}

__global__ __launch_bounds__(64) void minmax_final_kernel(const float *__restrict__ partial, int nb, float *__restrict__ out) {
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int i = threadIdx.x; i < nb; i += 64) { mn = fminf(mn, partial[2 * i]); mx = fmaxf(mx, partial[2 * i + 1]); }
    mn = -l2d_wave_max(-mn);
    mx = l2d_wave_max(mx);
    if (threadIdx.x == 0) { out[0] = mn; out[1] = mx; }
}

int l2d_launch_minmax(const l2d_op *op, hipStream_t s) {
    const long long n = op->l[0];
    const int nb = op->i[0];
    if (!op->p[0] || !op->p[1] || !op->p[2] || n <= 0 || nb <= 0 || nb > 1024 || (((uintptr_t)op->p[0]) & 15)) {
        l2d_set_error("minmax(tag %d): invalid arguments (n=%lld nb=%d)", op->tag, n, nb);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(256), 0, s, (const h16 *)op->p[0], n, (float *)op->p[1]);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, s, (const float *)op->p[1], nb, (float *)op->p[2]);
    return l2d_check_launch("minmax", op->tag);
}

// ---- (d - min) / (max - min) -> x 3 channels -> * 2 - 1 -> bilinear resize, in one pass
__global__ __launch_bounds__(256) void depth_norm_resize_kernel(const h16 *__restrict__ d, const float *__restrict__ mm,
                                                                h16 *__restrict__ out, int B, int Hd, int Wd, int H, int W) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W;
    if (idx >= total) return;
    const int x = (int)(idx % W);
    const long long r = idx / W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    // the reference's scalars are fp16 tensors: min, and (max - min) rounded to fp16
    const h16 mn = (h16)mm[0];
    const h16 range = (h16)((float)(h16)mm[1] - (float)mn);
    auto norm = [&](h16 v) -> float {       // fp16 rounding after each of: v - min, / range, * 2, - 1
        const h16 t1 = (h16)((float)v - (float)mn);
        const h16 t2 = (h16)((float)t1 / (float)range);
        const h16 t3 = (h16)((float)t2 * 2.0f);
        return (float)(h16)((float)t3 - 1.0f);
    };
    int y0, y1, x0, x1;
    float ly, lx;
    l2d_bilerp_coord(y, Hd, H, y0, y1, ly);
    l2d_bilerp_coord(x, Wd, W, x0, x1, lx);
    const h16 *p = d + (long long)b * Hd * Wd;
    const float v00 = norm(p[(long long)y0 * Wd + x0]), v01 = norm(p[(long long)y0 * Wd + x1]);
    const float v10 = norm(p[(long long)y1 * Wd + x0]), v11 = norm(p[(long long)y1 * Wd + x1]);
    const h16 o = (h16)((1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11));
    const long long plane = (long long)H * W;
    h16 *q = out + (long long)b * 3 * plane + (long long)y * W + x;
    q[0] = o; q[plane] = o; q[2 * plane] = o;
}

int l2d_launch_depth_norm_resize(const l2d_op *op, hipStream_t s) {
    const int B = op->i[0], Hd = op->i[1], Wd = op->i[2], H = op->i[3], W = op->i[4];
    if (!op->p[0] || !op->p[1] || !op->p[2] || B <= 0 || Hd <= 0 || Wd <= 0 || H <= 0 || W <= 0) {
        l2d_set_error("depth_norm_resize(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(depth_norm_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16 *)op->p[0],
                       (const float *)op->p[1], (h16 *)op->p[2], B, Hd, Wd, H, W);
    return l2d_check_launch("depth_norm_resize", op->tag);
}

// ================================================================================================================
// Small kernels of the depth detector (DPT-Hybrid; reference depth_utils.py:11-32 loads it from
// torch.hub "lewiji/MiDaS").  Its GEMM-shaped work runs on the igemm / flash kernels; these are the layout / pooling /
// elementwise pieces around them.  Activations are channels-last fp16.

// ResNetV2 stem: weight-standardised 7x7 stride-2 convolution with TF-"SAME" padding, 3 -> 64 channels, reading the NCHW
// image directly (the only 3-channel tensor of the network) and writing channels-last.  One thread = one output pixel x 8
// output channels; the 64 x 147 weights sit in LDS as [tap][channel] so a thread's 8 channels are one 16-byte read.
__global__ __launch_bounds__(256) void stem7x7_kernel(const h16 *__restrict__ img, const h16 *__restrict__ w, h16 *__restrict__ out, int B,
                                                      int H, int W, int Ho, int Wo, int pad_t, int pad_l) {
    __shared__ __attribute__((aligned(16))) h16 ws[147 * 64];
    for (int i = threadIdx.x; i < 147 * 64; i += 256) {
        const int co = i & 63, tap = i >> 6;                 // tap = ci*49 + ky*7 + kx ; packed weight is [co][ci][ky][kx]
        ws[i] = w[co * 147 + tap];
    }
    __syncthreads();
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * Ho * Wo * 8;
    if (idx >= total) return;
    const int cg = (int)(idx & 7);
    const long long pix = idx >> 3;
    const int ox = (int)(pix % Wo);
    const long long r = pix / Wo;
    const int oy = (int)(r % Ho), b = (int)(r / Ho);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int ci = 0; ci < 3; ++ci) {
        const h16 *pl = img + ((long long)b * 3 + ci) * H * W;
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = oy * 2 + ky - pad_t;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int ix = ox * 2 + kx - pad_l;
                if (ix < 0 || ix >= W) continue;
                const float x = (float)pl[(long long)iy * W + ix];
                const h16x8 wv = l2d_ld8(ws + (ci * 49 + ky * 7 + kx) * 64 + cg * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(x, (float)wv[e], acc[e]);
            }
        }
    }
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)acc[e];
    l2d_st8(out + pix * 64 + cg * 8, o);
}

int l2d_launch_stem7x7(const l2d_op *op, hipStream_t s) {
    const int B = op->i[0], H = op->i[1], W = op->i[2];
    if (!op->p[0] || !op->p[1] || !op->p[2] || B <= 0 || H <= 0 || W <= 0) {
        l2d_set_error("stem7x7(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int th = (Ho - 1) * 2 + 7 - H, tw = (Wo - 1) * 2 + 7 - W;
    const long long total = (long long)B * Ho * Wo * 8;
    hipLaunchKernelGGL(stem7x7_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16 *)op->p[0], (const h16 *)op->p[1],
                       (h16 *)op->p[2], B, H, W, Ho, Wo, (th > 0 ? th : 0) / 2, (tw > 0 ? tw : 0) / 2);
    return l2d_check_launch("stem7x7", op->tag);
}

// pooling / sampling on channels-last tensors, one thread per (output pixel, 8 channels):
//   mode 0: 3x3 stride-2 max pool with TF-"SAME" padding (window rows 2y-pad .. 2y-pad+2 that exist)
//   mode 1: stride-2 subsample (the 1x1 stride-2 shortcut convolution's gather)
//   mode 2: bilinear x2 upsample, align_corners=True (MiDaS Interpolate / FeatureFusionBlock)
__global__ __launch_bounds__(256) void resample_nhwc_kernel(const h16 *__restrict__ in, h16 *__restrict__ out, int B, int H, int W, int C,
                                                            int Ho, int Wo, int mode, int pad_t, int pad_l) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nvc = C / 8;
    const long long total = (long long)B * Ho * Wo * nvc;
    if (idx >= total) return;
    const int vc = (int)(idx % nvc);
    const long long pix = idx / nvc;
    const int ox = (int)(pix % Wo);
    const long long r = pix / Wo;
    const int oy = (int)(r % Ho), b = (int)(r / Ho);
    const h16 *src = in + (long long)b * H * W * C + vc * 8;
    h16x8 o;
    if (mode == 0) {
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 + ky - pad_t;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 + kx - pad_l;
                if (ix < 0 || ix >= W) continue;
                const h16x8 v = l2d_ld8(src + ((long long)iy * W + ix) * C);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)m[e];
    } else if (mode == 1) {
        o = l2d_ld8(src + ((long long)(oy * 2) * W + ox * 2) * C);
    } else {
        const float sy = Ho > 1 ? (float)oy * (float)(H - 1) / (float)(Ho - 1) : 0.f;
        const float sx = Wo > 1 ? (float)ox * (float)(W - 1) / (float)(Wo - 1) : 0.f;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const h16x8 v00 = l2d_ld8(src + ((long long)y0 * W + x0) * C), v01 = l2d_ld8(src + ((long long)y0 * W + x1) * C);
        const h16x8 v10 = l2d_ld8(src + ((long long)y1 * W + x0) * C), v11 = l2d_ld8(src + ((long long)y1 * W + x1) * C);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = (h16)((1.f - ly) * ((1.f - lx) * (float)v00[e] + lx * (float)v01[e]) + ly * ((1.f - lx) * (float)v10[e] + lx * (float)v11[e]));
    }
    l2d_st8(out + pix * C + vc * 8, o);
}

int l2d_launch_resample_nhwc(const l2d_op *op, hipStream_t s) {
    const int B = op->i[0], H = op->i[1], W = op->i[2], C = op->i[3], mode = op->i[4];
    if (!op->p[0] || !op->p[1] || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) || mode < 0 || mode > 2) {
        l2d_set_error("resample_nhwc(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    const int Ho = mode == 2 ? 2 * H : (H + 1) / 2, Wo = mode == 2 ? 2 * W : (W + 1) / 2;
    const int th = (Ho - 1) * 2 + 3 - H, tw = (Wo - 1) * 2 + 3 - W;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(resample_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16 *)op->p[0], (h16 *)op->p[1],
                       B, H, W, C, Ho, Wo, mode, mode == 0 ? (th > 0 ? th : 0) / 2 : 0, mode == 0 ? (tw > 0 ? tw : 0) / 2 : 0);
    return l2d_check_launch("resample_nhwc", op->tag);
}

// elementwise: s = a (+ b); out = s (if given); out_relu = relu(s) (if given).  The pre-activation residual conv units of the
// DPT decoder need a tensor and its ReLU at the same time.
__global__ __launch_bounds__(256) void ew_kernel(const h16 *__restrict__ a, const h16 *__restrict__ b, h16 *__restrict__ out,
                                                 h16 *__restrict__ out_relu, long long n8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    h16x8 v = l2d_ld8(a + i * 8);
    if (b) v = v + l2d_ld8(b + i * 8);
    if (out) l2d_st8(out + i * 8, v);
    if (out_relu) l2d_st8(out_relu + i * 8, __builtin_elementwise_max(v, l2d_zero8()));
}

int l2d_launch_ew(const l2d_op *op, hipStream_t s) {
    const long long n = op->l[0];
    if (!op->p[0] || (!op->p[2] && !op->p[3]) || n <= 0 || (n % 8)) {
        l2d_set_error("ew(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    hipLaunchKernelGGL(ew_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, (const h16 *)op->p[0], (const h16 *)op->p[1],
                       (h16 *)op->p[2], (h16 *)op->p[3], n / 8);
    return l2d_check_launch("ew", op->tag);
}
