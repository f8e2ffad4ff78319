// Small latency-bound kernels on the UNet boundary (gfx950):
//   timestep_embed : diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`
//                    (call site reference unet_depth_streaming.py:102,499)
//   skinny_linear  : nn.Linear for <= 8 rows (time embedding MLP + every resnet's time_emb_proj,
//                    reference unet_depth_streaming.py:505, resnet.py:238): one wave per output column,
//                    weights streamed once with 16-byte loads, wave-shuffle reduction
//   nchw<->nhwc    : the reference interface is NCFHW (f=1); the backend is channels-last end to end,
//                    so the only two layout conversions of the whole step are on the 4-channel latents
//   lcm_step       : scheduler_step_batch (reference pipeline_stream_animation_depth.py:387-401)
//   copy_bench     : float4 device copy used to quote the measured HBM peak next to the roofline
#include "common.h"

__global__ void timestep_embed_kernel(const long long *__restrict__ t, h16 *__restrict__ out, int N, int dim) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int half = dim / 2;
    if (idx >= N * half) return;
    int n = idx / half, i = idx - n * half;
    float freq = __expf(-9.210340371976184f * (float)i / (float)half);   // ln(10000)
    float ang = (float)t[n] * freq;
    float sn, cs;
    sincosf(ang, &sn, &cs);
    out[(long long)n * dim + i] = (h16)cs;          // flip_sin_to_cos: [cos | sin]
    out[(long long)n * dim + half + i] = (h16)sn;
}

int l2d_launch_timestep_embed(const l2d_op *op, hipStream_t s) {
    int N = op->i[0], dim = op->i[1];
    if (!op->p[0] || !op->p[1] || N <= 0 || dim <= 0 || (dim & 1)) {
        l2d_set_error("timestep_embed(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    int total = N * dim / 2;
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, s, (const long long *)op->p[0],
                       (h16 *)op->p[1], N, dim);
    return l2d_check_launch("timestep_embed", op->tag);
}

// out[m][n] = act( sum_k A[m][k] W[n][k] + b[n] ), M <= 8.  One wave per column n.
template <int MM>
__global__ __launch_bounds__(256) void skinny_linear_kernel(const h16 *__restrict__ A, const h16 *__restrict__ W,
                                                            const float *__restrict__ bias, void *__restrict__ out, int K,
                                                            int Nout, int silu_out, int out_is_float, int ldo, long long lda) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= Nout) return;
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    const int nvc = K / 8;
    for (int vc = lane; vc < nvc; vc += 64) {
        h16x8 w = l2d_ld8(W + (long long)n * K + vc * 8);
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            h16x8 x = l2d_ld8(A + (long long)m * lda + vc * 8);
            // v_dot2_f32_f16 for EVERY row, spelled out: left to the compiler (`acc += (float) w[e] * (float) x[e]`, fp-contract
            // fast) the unrolled rows of one thread got DIFFERENT instruction sequences -- row 0 a chain of v_fma_mix_f32, row 1
            // v_dot2c_f32_f16 (round 5: seen in the ISA once packed fp32 math was switched off) -- and with them different
            // roundings: the stream-batch rows of the time embedding stopped being interchangeable
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[m] = __builtin_amdgcn_fdot2((h16x2){w[2 * e], w[2 * e + 1]}, (h16x2){x[2 * e], x[2 * e + 1]}, acc[m], false);
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        float v = l2d_wave_sum(acc[m]);
        if (lane == 0) {
            if (bias) v += bias[n];
            if (silu_out) v = l2d_silu(v);
            if (out_is_float) ((float *)out)[(long long)m * ldo + n] = v;
            else ((h16 *)out)[(long long)m * ldo + n] = (h16)v;
        }
    }
}

int l2d_launch_skinny_linear(const l2d_op *op, hipStream_t s) {
    const h16 *A = (const h16 *)op->p[0], *W = (const h16 *)op->p[1];
    const float *bias = (const float *)op->p[2];
    void *out = op->p[3];
    int M = op->i[0], K = op->i[1], Nout = op->i[2], silu = op->i[3], isf = op->i[4], ldo = op->i[5];
    const long long lda = op->l[0] > 0 ? op->l[0] : K;       // row stride of A in halfs (0: dense)
    if (!A || !W || !out || M <= 0 || M > 8 || K <= 0 || (K % 8) || Nout <= 0 || ldo < Nout || (lda % 8)) {
        l2d_set_error("skinny_linear(tag %d): invalid arguments (M=%d K=%d N=%d)", op->tag, M, K, Nout);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    dim3 grid((Nout + 3) / 4), block(256);
#define L2D_SK(MMV) \
    case MMV: hipLaunchKernelGGL((skinny_linear_kernel<MMV>), grid, block, 0, s, A, W, bias, out, K, Nout, silu, isf, ldo, lda); break;
    switch (M) {
        L2D_SK(1) L2D_SK(2) L2D_SK(3) L2D_SK(4) L2D_SK(5) L2D_SK(6) L2D_SK(7) L2D_SK(8)
    }
#undef L2D_SK
    return l2d_check_launch("skinny_linear", op->tag);
}

// mode 0: copy; 1: (x + b) * a (TAESD encoder input, x.add(1).div(2)); 2: tanh(x / 3) * 3 (TAESD decoder input);
// 3: x * a + b (TAESD decoder output, x.mul(2).sub(1)).  Each step rounds to fp16 like the reference's fp16 tensor ops.
__device__ __forceinline__ h16 l2d_layout_map(h16 x, int mode, float a, float b) {
    if (mode == 1) return (h16)((float)(h16)((float)x + b) * a);
    if (mode == 2) return (h16)((float)(h16)tanhf((float)(h16)((float)x * (1.0f / 3.0f))) * 3.0f);
    if (mode == 3) return (h16)((float)(h16)((float)x * a) + b);
    return x;
}

// One thread per output pixel: Cpad (8) halfs = one 16-byte store; the C plane reads are coalesced across threads.
__global__ void nchw_to_nhwc_kernel(const h16 *__restrict__ in, h16 *__restrict__ out, int B, int C, int HW, int Cpad, int mode,
                                    float a, float b) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * HW * Cpad;
    if (idx >= total) return;
    int c = (int)(idx % Cpad);
    long long pix = idx / Cpad;
    int bb = (int)(pix / HW);
    int hw = (int)(pix - (long long)bb * HW);
    out[idx] = (c < C) ? l2d_layout_map(in[((long long)bb * C + c) * HW + hw], mode, a, b) : (h16)0.0f;
}

__global__ void nhwc_to_nchw_kernel(const h16 *__restrict__ in, h16 *__restrict__ out, int B, int C, int HW, int ld, int mode, float a,
                                    float b) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)B * C * HW;
    if (idx >= total) return;
    int hw = (int)(idx % HW);
    long long bc = idx / HW;
    int c = (int)(bc % C);
    int bb = (int)(bc / C);
    out[idx] = l2d_layout_map(in[((long long)bb * HW + hw) * ld + c], mode, a, b);
}

int l2d_launch_nchw_to_nhwc(const l2d_op *op, hipStream_t s) {
    int B = op->i[0], C = op->i[1], HW = op->i[2], Cpad = op->i[3], mode = op->i[4];
    if (!op->p[0] || !op->p[1] || B <= 0 || C <= 0 || HW <= 0 || Cpad < C || mode < 0 || mode > 3) {
        l2d_set_error("nchw_to_nhwc(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    long long total = (long long)B * HW * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16 *)op->p[0],
                       (h16 *)op->p[1], B, C, HW, Cpad, mode, op->f[0], op->f[1]);
    return l2d_check_launch("nchw_to_nhwc", op->tag);
}

int l2d_launch_nhwc_to_nchw(const l2d_op *op, hipStream_t s) {
    int B = op->i[0], C = op->i[1], HW = op->i[2], ld = op->i[3], mode = op->i[4];
    if (!op->p[0] || !op->p[1] || B <= 0 || C <= 0 || HW <= 0 || ld < C || mode < 0 || mode > 3) {
        l2d_set_error("nhwc_to_nchw(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    long long total = (long long)B * C * HW;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16 *)op->p[0],
                       (h16 *)op->p[1], B, C, HW, ld, mode, op->f[0], op->f[1]);
    return l2d_check_launch("nhwc_to_nchw", op->tag);
}

// x0 = c_out * (x - beta*eps)/alpha + c_skip * x ; scal[n] = {alpha, beta, c_skip, c_out}
__global__ void lcm_step_kernel(const h16 *__restrict__ x, const h16 *__restrict__ eps, const float *__restrict__ scal,
                                h16 *__restrict__ x0, int N, int per) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * per) return;
    int n = (int)(idx / per);
    float al = scal[n * 4 + 0], be = scal[n * 4 + 1], cs = scal[n * 4 + 2], co = scal[n * 4 + 3];
    float xv = (float)x[idx], ev = (float)eps[idx];
    // same fp16 rounding points as the reference's half-precision tensor expression (:395-396)
    h16 f = (h16)((float)(h16)(xv - (float)(h16)(be * ev)) / al);
    x0[idx] = (h16)((float)(h16)(co * (float)f) + (float)(h16)(cs * xv));
}

int l2d_launch_lcm_step(const l2d_op *op, hipStream_t s) {
    int N = op->i[0], per = op->i[1];
    if (!op->p[0] || !op->p[1] || !op->p[2] || !op->p[3] || N <= 0 || per <= 0) {
        l2d_set_error("lcm_step(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    long long total = (long long)N * per;
    hipLaunchKernelGGL(lcm_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16 *)op->p[0],
                       (const h16 *)op->p[1], (const float *)op->p[2], (h16 *)op->p[3], N, per);
    return l2d_check_launch("lcm_step", op->tag);
}

// ------------------------------------------------------------------------------------------- HBM copy probe
// float4 copy: 4 independent 16-byte loads in flight per thread, non-temporal stores (the written lines are not read again:
// keeping them out of the L2 / Infinity Cache leaves those to the read stream)
__global__ __launch_bounds__(256) void copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, long long n16) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride);
        __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int l2d_copy_bench(const void *src, void *dst, int64_t bytes, int reps, void *stream, float *gbps_out) {
    if (!src || !dst || bytes < 16 || reps <= 0 || !gbps_out) {
        l2d_set_error("copy_bench: invalid arguments");
        return L2D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    long long n16 = bytes / 16;
    int grid = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s, (const f32x4 *)src, (f32x4 *)dst, n16);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s, (const f32x4 *)src, (f32x4 *)dst, n16);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *gbps_out = (float)(2.0 * (double)n16 * 16.0 * reps / (ms * 1e-3) / 1e9);
    return l2d_check_launch("copy_bench", 0);
}

// ------------------------------------------------------------------------------------------- HBM read probe
// Pure streaming read (xor-reduce so the loads cannot be dropped), U independent 16-byte loads in flight per
// thread, blocks_per_cu resident blocks: measures what a read-dominated kernel such as the KV-cache attention
// can expect from this part (the copy probe above spends half of its traffic on writes).
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const f32x4 *__restrict__ src, unsigned *__restrict__ sink, long long n16) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x);
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= __float_as_uint(v[u][0]) ^ __float_as_uint(v[u][1]) ^ __float_as_uint(v[u][2]) ^ __float_as_uint(v[u][3]);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int l2d_read_bench(const void *src, void *sink, int64_t bytes, int unroll, int blocks_per_cu, int reps, void *stream,
                              float *gbps_out) {
    if (!src || !sink || bytes < 16 || reps <= 0 || !gbps_out || blocks_per_cu <= 0) {
        l2d_set_error("read_bench: invalid arguments");
        return L2D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    long long n16 = bytes / 16;
    int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        switch (unroll) {
            case 1: hipLaunchKernelGGL((read_kernel<1>), dim3(grid), dim3(256), 0, s, (const f32x4 *)src, (unsigned *)sink, n16); break;
            case 2: hipLaunchKernelGGL((read_kernel<2>), dim3(grid), dim3(256), 0, s, (const f32x4 *)src, (unsigned *)sink, n16); break;
            case 4: hipLaunchKernelGGL((read_kernel<4>), dim3(grid), dim3(256), 0, s, (const f32x4 *)src, (unsigned *)sink, n16); break;
            case 8: hipLaunchKernelGGL((read_kernel<8>), dim3(grid), dim3(256), 0, s, (const f32x4 *)src, (unsigned *)sink, n16); break;
            default: hipLaunchKernelGGL((read_kernel<16>), dim3(grid), dim3(256), 0, s, (const f32x4 *)src, (unsigned *)sink, n16); break;
        }
    };
    launch();
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *gbps_out = (float)((double)n16 * 16.0 * reps / (ms * 1e-3) / 1e9);
    return l2d_check_launch("read_bench", 0);
}

#ifdef L2D_PROBES
// Per-CU ingest probe (analysis builds only, tools/ingest_probe.py): `grid` blocks (one per CU while grid <= 256) of `waves`
// waves; every wave keeps U wave-instructions of 1 KB (16 B per lane) in flight, `iters` batches.  mode 0: every wave streams its
// private region (cold: HBM), mode 1: every wave walks the SAME `region` bytes (L2 / Infinity-Cache hits), mode 2 / 3: the same
// two patterns through global_load_lds (LDS-DMA, no VGPR staging).  Answers: what can ONE CU pull from HBM / from L2, with how
// many requests in flight, and how does that scale with the number of active CUs.
template <int U, int MODE>
__global__ __launch_bounds__(1024) void ingest_kernel(const char *src, unsigned *sink, long long region, int iters) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const long long gw = (long long)blockIdx.x * nw + wave;
    const bool priv = (MODE == 0 || MODE == 2);
    const char *base = priv ? src + gw * region : src + (gw * 4096) % region;
    const long long wrap = region - U * 1024;
    unsigned acc = 0;
    long long off = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)(base + off + u * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(ring + (wave * U + u) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4 *>(base + off + u * 1024 + lane * 16);
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= __float_as_uint(v[u][0]) ^ __float_as_uint(v[u][3]);
        }
        off += U * 1024;
        if (off > wrap) off = 0;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int l2d_ingest_bench(const void *src, void *sink, int64_t region, int mode, int grid, int waves, int inflight, int iters,
                                void *stream, float *gbps_out) {
    if (!src || !sink || !gbps_out || grid <= 0 || waves <= 0 || waves > 16 || iters <= 0 || region < inflight * 1024) {
        l2d_set_error("ingest_bench: invalid arguments");
        return L2D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = mode >= 2 ? (size_t)waves * inflight * 1024 : 0;
    auto launch = [&]() {
#define ING(U, M) hipLaunchKernelGGL((ingest_kernel<U, M>), dim3(grid), dim3(64 * waves), lds, s, (const char *)src, (unsigned *)sink, (long long)region, iters)
#define INGU(M) switch (inflight) { case 2: ING(2, M); break; case 4: ING(4, M); break; case 8: ING(8, M); break; case 16: ING(16, M); break; default: ING(32, M); break; }
        switch (mode) { case 0: INGU(0) break; case 1: INGU(1) break; case 2: INGU(2) break; default: INGU(3) break; }
#undef INGU
#undef ING
    };
    if (lds > 65536) {
        l2d_set_error("ingest_bench: LDS-DMA modes take waves * inflight <= 64");
        return L2D_EINVAL;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipEventRecord(e0, s);
    launch();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *gbps_out = (float)((double)grid * waves * inflight * 1024.0 * iters / (ms * 1e-3) / 1e9);
    return l2d_check_launch("ingest_bench", 0);
}

// Transcendental-issue probe (analysis builds only, tools/exp_probe.py): how many cycles does a wave64 v_exp_f32 cost on gfx950 --
// alone, with a second wave on the SIMD, and beside dependent MFMAs?  Settles the VALU floor of the flash-attention softmax
// (one exponential per score).  MODE 0: 16 independent v_exp_f32 per iteration; 1: 16 independent
// v_add_f32 (the plain-VALU yardstick); 2: 16 v_exp_f32 + 14 v_mfma_f32_16x16x32_f16 on 2 accumulators (the ratio of one 64-key
// tile of the d = 40 kernel: 32 exponentials to 28 MFMAs per wave); 3: the 14 MFMAs alone; 4: 16 v_add + 14 MFMAs.
// One block per CU, `waves` per block (4 = one per SIMD, 8 = two per SIMD); out[wave] = cycles for `iters` iterations.
typedef float f32x4p __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void exp_probe_kernel(unsigned long long *out, int iters, float seed) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed * (float)(i + 1 + (threadIdx.x & 7)) * 1e-3f;
    f32x4p acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    typedef _Float16 hh8 __attribute__((ext_vector_type(8)));
    hh8 fa, fb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa[e] = (_Float16)(0.01f * (float)(e + 1)); fb[e] = (_Float16)(0.02f * (float)(e + 1)); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (MODE == 2 && i < 14) {
                    if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc0, 0, 0, 0);
                }
            }
        } else if (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
                if (MODE == 4 && i < 14) {
                    if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc0, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc0, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sink = acc0[0] + acc1[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) sink += x[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (sink == 1234.5f) ? 0ull : (t1 - t0);
}

extern "C" int l2d_exp_probe(void *out, int mode, int waves, int iters, void *stream) {
    if (!out || waves < 1 || waves > 8 || iters <= 0) {
        l2d_set_error("exp_probe: invalid arguments");
        return L2D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case 0: hipLaunchKernelGGL((exp_probe_kernel<0>), dim3(256), dim3(64 * waves), 0, s, (unsigned long long *)out, iters, 0.37f); break;
        case 1: hipLaunchKernelGGL((exp_probe_kernel<1>), dim3(256), dim3(64 * waves), 0, s, (unsigned long long *)out, iters, 0.37f); break;
        case 2: hipLaunchKernelGGL((exp_probe_kernel<2>), dim3(256), dim3(64 * waves), 0, s, (unsigned long long *)out, iters, 0.37f); break;
        case 3: hipLaunchKernelGGL((exp_probe_kernel<3>), dim3(256), dim3(64 * waves), 0, s, (unsigned long long *)out, iters, 0.37f); break;
        default: hipLaunchKernelGGL((exp_probe_kernel<4>), dim3(256), dim3(64 * waves), 0, s, (unsigned long long *)out, iters, 0.37f); break;
    }
    return l2d_check_launch("exp_probe", 0);
}

// Reproducer attempt for the round-4 packed-fp32 finding (tools/pk_repro.py): the LayerNorm-fold arithmetic of
// wsgemm.hip's epilogue in isolation -- per lane out[e] = bias[e] + (acc[e] * rstd + colsum[e] * nmr), once on 2-vectors (hipcc emits
// v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32, as in the kernel before the fix) and once element by element behind opaque barriers
// (scalar v_mul / v_fma / v_add, the same rounding steps) -- compared bit for bit in the kernel.  Geometry as the kernel's: four
// working waves with the parameters in LDS (bias | colsum as ds_read_b128, rstd | nmr per token as ds_read2_b32), a staged fp16
// tile (ds_write_b64), and two more waves that sleep at the final barrier on the SIMDs of waves 0 and 1.
typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef _Float16 h16x4p __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(384) void pk_repro_kernel(const float *par_g, const float *stat_g, unsigned int *nbad, unsigned int *first, int iters) {
    __shared__ __attribute__((aligned(16))) float par[512];
    __shared__ __attribute__((aligned(16))) float stat[256];
    __shared__ __attribute__((aligned(16))) _Float16 os[128 * 136];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 512; i += blockDim.x) par[i] = par_g[i];
    for (int i = tid; i < 256; i += blockDim.x) stat[i] = stat_g[i];
    __syncthreads();
    if (wave >= 4) { __syncthreads(); return; }
    float acc[4][16];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = (float)((lane * 37 + mt * 11 + r * 5 + (int)blockIdx.x) % 97) * 0.03125f - 1.5f;
    unsigned int bad = 0, where = 0xffffffffu;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int ch = wave * 32 + 8 * g4 + 4 * lh;
            const f32x4p bb = *reinterpret_cast<const f32x4p *>(par + ch), cs = *reinterpret_cast<const f32x4p *>(par + 256 + ch);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                float rsd = stat[(32 * mt + l32) * 2], nmr = stat[(32 * mt + l32) * 2 + 1];
                // MODE 1: idle cycles between the LDS wait and the first packed instruction that reads the loaded registers;
                // MODE 2: one plain VALU copy of the loaded values in between (the packed instructions read the copies)
                if constexpr (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(rsd), "+v"(nmr));
                if constexpr (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(rsd), "+v"(nmr));
                h16x4p o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2p c = {cs[2 * h], cs[2 * h + 1]}, a = {acc[mt][4 * g4 + 2 * h], acc[mt][4 * g4 + 2 * h + 1]};
                    const f32x2p b = {bb[2 * h], bb[2 * h + 1]};
                    f32x2p p = c * nmr;
                    asm volatile("" : "+v"(p));
                    p = a * rsd + p;
                    asm volatile("" : "+v"(p));
                    const f32x2p v = b + p;
                    const float vv[2] = {v.x, v.y};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float r = cs[2 * h + e] * nmr;
                        asm volatile("" : "+v"(r));
                        r = __builtin_fmaf(acc[mt][4 * g4 + 2 * h + e], rsd, r);
                        asm volatile("" : "+v"(r));
                        r = bb[2 * h + e] + r;
                        asm volatile("" : "+v"(r));
                        if (__builtin_bit_cast(unsigned int, r) != __builtin_bit_cast(unsigned int, vv[e])) {
                            ++bad;
                            if (where == 0xffffffffu) where = (unsigned int)((lane << 16) | (mt << 12) | (g4 << 8) | ((2 * h + e) << 4) | (wave));
                        }
                        o[2 * h + e] = (_Float16)vv[e];
                    }
                }
                *reinterpret_cast<h16x4p *>(os + (32 * mt + l32) * 136 + wave * 32 + 8 * g4 + 4 * lh) = o;
            }
        }
        // perturb the accumulators so that iterations differ (exactly representable steps)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = acc[mt][r] * -1.0f + 0.0078125f * (float)((it + r) & 3);
    }
    if (bad) {
        atomicAdd(nbad, bad);
        const unsigned int slot = atomicAdd(nbad + 1, 1u);
        if (slot < 16) { first[2 * slot] = where; first[2 * slot + 1] = blockIdx.x; }
    }
    if (os[tid] == (_Float16)12345.0f) nbad[2] = 1;          // (keeps the staged tile alive)
    __syncthreads();
}

extern "C" int l2d_pk_repro(const void *par, const void *stat, void *nbad, void *first, int blocks, int iters, void *stream) {
    if (!par || !stat || !nbad || !first || blocks <= 0 || iters <= 0) {
        l2d_set_error("pk_repro: invalid arguments");
        return L2D_EINVAL;
    }
    const int mode = blocks >> 16;
    blocks &= 0xffff;
    if (mode == 1)
        hipLaunchKernelGGL(pk_repro_kernel<1>, dim3(blocks), dim3(384), 0, (hipStream_t)stream, (const float *)par, (const float *)stat,
                           (unsigned int *)nbad, (unsigned int *)first, iters);
    else if (mode == 2)
        hipLaunchKernelGGL(pk_repro_kernel<2>, dim3(blocks), dim3(384), 0, (hipStream_t)stream, (const float *)par, (const float *)stat,
                           (unsigned int *)nbad, (unsigned int *)first, iters);
    else
        hipLaunchKernelGGL(pk_repro_kernel<0>, dim3(blocks), dim3(384), 0, (hipStream_t)stream, (const float *)par, (const float *)stat,
                           (unsigned int *)nbad, (unsigned int *)first, iters);
    return l2d_check_launch("pk_repro", 0);
}

// Access-pattern probe for the KV-cache stream (analysis builds only, tools/kv_pattern_probe.py).  One 320-thread block per CU
// streams its private slice of `src` HBM -> LDS with the ring discipline of tattn_ring.hip (NS stages of R DMA wave-instructions,
// counted vmcnt waits, one barrier per stage) and NO arithmetic.  `pattern` picks what a stage fetches from a group of 8 pixels
// x 16 rows x 640 B (= 80 KB contiguous, the [pixel][slot][channel] slab of the reference layout at C = 320):
//   0  rows 4s .. 4s+3 of all 8 pixels: 8 pieces of 2.5 KB at 10 KB stride (what tattn_ring.hip does)
//   1  all 16 rows of pixels 2s, 2s+1: one contiguous 20 KB run
//   2  the group read front to back, 20 KB per stage (the same bytes as 1; source order = LDS order)
// K and V slabs are `slab` bytes apart, visited K (4 stages) then V (4 stages) per group, like the real kernel.
template <int NS>
__global__ __launch_bounds__(320) void kv_pattern_kernel(const char *src, unsigned *sink, long long slab, int groups_per_block, int pattern) {
    constexpr int R = 4, STAGE = R * 320 * 16;
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = tid / 40, cc = tid - p * 40;
    const char *base = src + (long long)blockIdx.x * groups_per_block * 81920;
    const int total = groups_per_block * 8;
    int it_issue = 0, slot = 0;
    auto issue = [&]() {
        const int g = it_issue >> 3, s = it_issue & 7, sk = s & 3;
        const char *gb = base + (long long)g * 81920 + (s >= 4 ? slab : 0);
        char *dst = ring + slot * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            long long off;
            if (pattern == 0) off = (long long)p * 10240 + (sk * 4 + j) * 640 + cc * 16;
            else off = (long long)sk * 20480 + (j * 320 + tid) * 16;
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)(gb + off),
                                             (__attribute__((address_space(3))) void *)(dst + j * 320 * 16), 16, 0, 0);
        }
        ++it_issue;
        slot = (slot + 1 == NS) ? 0 : slot + 1;
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (it_issue < total) issue();
    unsigned acc = 0;
    int cs = 0;
    for (int it = 0; it < total; ++it) {
        if (total - 1 - it >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * R) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it_issue < total) issue();
        acc ^= *reinterpret_cast<const unsigned *>(ring + cs * STAGE + tid * 16);      // one LDS read per stage keeps the data "used"
        cs = (cs + 1 == NS) ? 0 : cs + 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int l2d_kv_pattern_bench(const void *src, void *sink, int64_t slab, int groups_per_block, int pattern, int blocks, int reps,
                                    void *stream, float *gbps_out) {
    hipStream_t s = (hipStream_t)stream;
    constexpr int NS = 5;
    const int lds = NS * 4 * 320 * 16;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void *)kv_pattern_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() { hipLaunchKernelGGL((kv_pattern_kernel<NS>), dim3(blocks), dim3(320), lds, s, (const char *)src, (unsigned *)sink, (long long)slab, groups_per_block, pattern); };
    launch();
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *gbps_out = (float)((double)blocks * groups_per_block * 2.0 * 81920.0 * reps / (ms * 1e-3) / 1e9);
    return l2d_check_launch("kv_pattern_bench", 0);
}
#endif
