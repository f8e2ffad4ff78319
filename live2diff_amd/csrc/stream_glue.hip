// Per-frame glue of the streaming pipeline on the device: what the reference does
// between two UNet calls with Python loops, `.any()` / `.sum()` host syncs and a dozen tiny tensor ops.
//
//   ring_update   : StreamAnimateDiffusionDepth.update_attn_bias (reference pipeline_stream_animation_depth.py:416-438):
//                   the ring-buffer state machine over (attn_bias [N,L], pe_idx [N,L], update_idx [N]), in place on
//                   the UNet's static input buffers -- no host round trip, no H2D copies.
//   stream_shift  : scheduler_step_batch (:387-401) + the stream-batch shift register of predict_x0_batch (:590-601):
//                   x0 = c_out (x - beta eps) / alpha + c_skip x for the N rows, output = x0 of the last row,
//                   next buffer row i+1 = alpha[i+1] x0[i] + beta[i+1] noise[i], depth row i+1 = depth row i;
//                   written straight into the UNet's static input buffers (same fp16 rounding points as the
//                   reference's half-precision tensor expressions).
//   randn         : the re-noising tensor (reference: torch.randn_like, :596-598).  Counter-based Philox4x32-10 +
//                   Box-Muller; the reference's generator stream is backend specific, so parity is distributional.
#include "common.h"

// ------------------------------------------------------------------------------------------- ring buffer
__global__ void ring_update_kernel(h16 *__restrict__ bias, long long *__restrict__ pe_idx, long long *__restrict__ upd,
                                   unsigned long long *__restrict__ frame_ctr, int N, int L, int sink) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && frame_ctr) *frame_ctr += 1;         // frame counter of the device-side pipeline (noise stream position)
    if (n >= N) return;
    h16 *b = bias + (long long)n * L;
    long long *p = pe_idx + (long long)n * L;
    int filled = 0;
    for (int l = 0; l < L; ++l) filled += ((float)b[l] == 0.0f) ? 1 : 0;
    if (filled < L) {
        upd[n] = filled;                              // some slot still masked: append (:425-427)
    } else {
        // full: rotate the rolling part's positional indices right by one, overwrite the slot that now holds the
        // largest index (first maximum, as torch.argmax) (:428-434)
        const long long last = p[L - 1];
        for (int l = L - 1; l > sink; --l) p[l] = p[l - 1];
        p[sink] = last;
        int arg = 0;
        long long mx = p[0];
        for (int l = 1; l < L; ++l)
            if (p[l] > mx) { mx = p[l]; arg = l; }
        upd[n] = arg;
    }
    const int upto = min(filled + 1, L);              // unmask one more slot (:436)
    for (int l = 0; l < upto; ++l) b[l] = (h16)0.0f;
}

int l2d_launch_ring_update(const l2d_op *op, hipStream_t s) {
    const int N = op->i[0], L = op->i[1], sink = op->i[2];
    if (!op->p[0] || !op->p[1] || !op->p[2] || N <= 0 || L <= 0 || sink < 0 || sink >= L) {
        l2d_set_error("ring_update(tag %d): invalid arguments (N=%d L=%d sink=%d)", op->tag, N, L, sink);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    hipLaunchKernelGGL(ring_update_kernel, dim3((N + 63) / 64), dim3(64), 0, s, (h16 *)op->p[0], (long long *)op->p[1],
                       (long long *)op->p[2], (unsigned long long *)op->p[3], N, L, sink);
    return l2d_check_launch("ring_update", op->tag);
}

// ------------------------------------------------------------------------------------------- LCM step + shift register
constexpr int SHIFT_MAX_N = 8;

__global__ __launch_bounds__(256) void stream_shift_kernel(h16 *__restrict__ x, const h16 *__restrict__ eps,
                                                           const float *__restrict__ scal, const h16 *__restrict__ noise,
                                                           h16 *__restrict__ x0_out, h16 *__restrict__ depth, int N, int per) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= per) return;
    h16 x0[SHIFT_MAX_N], dv[SHIFT_MAX_N];
#pragma unroll
    for (int n = 0; n < SHIFT_MAX_N; ++n) {
        if (n >= N) break;
        const float al = scal[n * 4 + 0], be = scal[n * 4 + 1], cs = scal[n * 4 + 2], co = scal[n * 4 + 3];
        const float xv = (float)x[(long long)n * per + e], ev = (float)eps[(long long)n * per + e];
        // fp16 rounding after every tensor op of the reference expression (:395-396), as in lcm_step_kernel
        const h16 f = (h16)((float)(h16)(xv - (float)(h16)(be * ev)) / al);
        x0[n] = (h16)((float)(h16)(co * (float)f) + (float)(h16)(cs * xv));
        if (depth) dv[n] = depth[(long long)n * per + e];
    }
    x0_out[e] = x0[N - 1];
#pragma unroll
    for (int n = 0; n + 1 < SHIFT_MAX_N; ++n) {
        if (n + 1 >= N) break;
        const float al = scal[(n + 1) * 4 + 0], be = scal[(n + 1) * 4 + 1];
        h16 v = (h16)(al * (float)x0[n]);                                        // alpha_prod_t_sqrt[1:] * x0[:-1]
        if (noise) v = (h16)((float)v + (float)(h16)(be * (float)noise[(long long)n * per + e]));   // + beta[1:] * noise
        x[(long long)(n + 1) * per + e] = v;
        if (depth) depth[(long long)(n + 1) * per + e] = dv[n];
    }
}

int l2d_launch_stream_shift(const l2d_op *op, hipStream_t s) {
    const int N = op->i[0], per = op->i[1];
    if (!op->p[0] || !op->p[1] || !op->p[2] || !op->p[4] || N <= 0 || N > SHIFT_MAX_N || per <= 0) {
        l2d_set_error("stream_shift(tag %d): invalid arguments (N=%d (max %d) per=%d)", op->tag, N, SHIFT_MAX_N, per);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    hipLaunchKernelGGL(stream_shift_kernel, dim3((per + 255) / 256), dim3(256), 0, s, (h16 *)op->p[0], (const h16 *)op->p[1],
                       (const float *)op->p[2], (const h16 *)op->p[3], (h16 *)op->p[4], (h16 *)op->p[5], N, per);
    return l2d_check_launch("stream_shift", op->tag);
}

// ------------------------------------------------------------------------------------------- normal noise
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// element i of the stream (seed, offset) is normal #(i % 4) of Philox block (offset + i / 4): position-wise reproducible
__global__ __launch_bounds__(256) void randn_kernel(h16 *__restrict__ out, long long n, unsigned long long seed,
                                                    unsigned long long offset, const unsigned long long *__restrict__ frame_ctr) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one Philox block = 4 normals
    if (q * 4 >= n) return;
    // a static plan (hipGraph) draws fresh noise every frame: the stream position advances by one tensor per frame
    const unsigned long long frame = frame_ctr ? *frame_ctr : 0ull;
    const unsigned long long ctr = offset + frame * (unsigned long long)((n + 3) / 4) + (unsigned long long)q;
    unsigned r[4];
    philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(r[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0, 1)
        const float u2 = ((float)(r[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        z[2 * h] = rad * cs;
        z[2 * h + 1] = rad * sn;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (q * 4 + j < n) out[q * 4 + j] = (h16)z[j];
}

int l2d_launch_randn(const l2d_op *op, hipStream_t s) {
    const long long n = op->l[0];
    if (!op->p[0] || n <= 0) {
        l2d_set_error("randn(tag %d): invalid arguments", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    const long long blocks = ((n + 3) / 4 + 255) / 256;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (h16 *)op->p[0], n, (unsigned long long)op->l[1],
                       (unsigned long long)op->l[2], (const unsigned long long *)op->p[1]);
    return l2d_check_launch("randn", op->tag);
}
