// Shared device/host helpers for the gfx950 kernels of libl2d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/l2d.h"

typedef _Float16 h16;
typedef h16 __attribute__((ext_vector_type(2))) h16x2;
typedef h16 __attribute__((ext_vector_type(4))) h16x4;
typedef h16 __attribute__((ext_vector_type(8))) h16x8;
typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef float __attribute__((ext_vector_type(2))) f32x2;

#define L2D_WAVE 64

// host-side error plumbing (capi.cpp)
void l2d_set_error(const char *fmt, ...);
int l2d_check_launch(const char *what, int tag);
// validate-only mode (l2d_set_dry_run): launchers return right after argument validation
extern int l2d_g_dry_run;
#define L2D_DRY_RETURN() do { if (l2d_g_dry_run) return L2D_OK; } while (0)

// > 64 KB of dynamic LDS must be opted into per kernel AND per device: launchers keep one flag per device ordinal
// (`static bool done[L2D_MAX_DEV]`; a process that drives several GPUs sets the attribute on each of them)
#define L2D_MAX_DEV 16
static inline int l2d_dev_ordinal() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= L2D_MAX_DEV) return 0;
    return dev;
}

// per-kernel launchers (one translation unit each); all return L2D_OK / L2D_E*
int l2d_launch_igemm(const l2d_op *op, hipStream_t s);
int l2d_launch_gn_stats(const l2d_op *op, hipStream_t s);
int l2d_launch_gn_apply(const l2d_op *op, hipStream_t s);
int l2d_launch_layernorm(const l2d_op *op, hipStream_t s);
int l2d_launch_flash_attn(const l2d_op *op, hipStream_t s);
int l2d_launch_tattn_stream(const l2d_op *op, hipStream_t s);
int l2d_launch_tattn_warmup(const l2d_op *op, hipStream_t s);
int l2d_launch_skinny_linear(const l2d_op *op, hipStream_t s);
int l2d_launch_timestep_embed(const l2d_op *op, hipStream_t s);
int l2d_launch_nchw_to_nhwc(const l2d_op *op, hipStream_t s);
int l2d_launch_nhwc_to_nchw(const l2d_op *op, hipStream_t s);
int l2d_launch_lcm_step(const l2d_op *op, hipStream_t s);
int l2d_launch_ring_update(const l2d_op *op, hipStream_t s);
int l2d_launch_stream_shift(const l2d_op *op, hipStream_t s);
int l2d_launch_randn(const l2d_op *op, hipStream_t s);
int l2d_launch_resize_bilinear(const l2d_op *op, hipStream_t s);
int l2d_launch_minmax(const l2d_op *op, hipStream_t s);
int l2d_launch_depth_norm_resize(const l2d_op *op, hipStream_t s);
int l2d_launch_stem7x7(const l2d_op *op, hipStream_t s);
int l2d_launch_resample_nhwc(const l2d_op *op, hipStream_t s);
int l2d_launch_ew(const l2d_op *op, hipStream_t s);
int l2d_launch_rowgemm(const l2d_op *op, hipStream_t s);
int l2d_launch_pconv(const l2d_op *op, hipStream_t s);
int l2d_launch_wsgemm(const l2d_op *op, hipStream_t s);
int l2d_launch_rowchain(const l2d_op *op, hipStream_t s);
int l2d_launch_cconv(const l2d_op *op, hipStream_t s);

#ifdef __HIPCC__
// SiLU / GELU are evaluated per output element inside GEMM epilogues and the GroupNorm apply pass (tens of millions of
// evaluations per frame), so they are built from single hardware instructions: v_exp_f32 (2^x) and v_rcp_f32 (1 ulp).
__device__ __forceinline__ float l2d_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// exact-erf GELU (torch.nn.functional.gelu default, used by diffusers' GEGLU) with erf from Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the fp16 rounding of the result).  The normal CDF is formed without cancellation:
// Phi(x) = 1 - 0.5 * tail(|x|) for x >= 0 and 0.5 * tail(|x|) for x < 0, tail(z) = erfc(z / sqrt 2).
__device__ __forceinline__ float l2d_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
    const float half_tail = 0.5f * poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    return x * (x >= 0.f ? 1.0f - half_tail : half_tail);
}

__device__ __forceinline__ h16x8 l2d_ld8(const h16 *p) { return *reinterpret_cast<const h16x8 *>(p); }
__device__ __forceinline__ void l2d_st8(h16 *p, h16x8 v) { *reinterpret_cast<h16x8 *>(p) = v; }
__device__ __forceinline__ h16x8 l2d_zero8() {
    h16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (h16)0.0f;
    return z;
}
__device__ __forceinline__ float l2d_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum x in units of 2^-20, sum x^2 in units of 2^-12: integer adds commute, so the accumulated statistics do not depend on
// the order in which blocks arrive (bit-repeatable frames), and |x| <= 65504 cannot overflow int64 at any size used here
#define L2D_GN_S1_SCALE 1048576.0f
#define L2D_GN_S2_SCALE 4096.0f

// chs: per-channel (sum, sum of squares) of this block's tile, channels [c_lo, c_lo + nch) of the producing tensor, all of
// sample `b`.  One thread per consumer group that overlaps the tile adds its channels and issues two integer atomics.
__device__ __forceinline__ void l2d_gn_flush(unsigned long long *acc, int G, int cpg, int choff, int b, const float *chs1,
                                               const float *chs2, int c_lo, int nch, int tid) {
    if (!acc || nch <= 0) return;
    const int first = choff + c_lo, last = first + nch - 1;
    const int g0 = first / cpg, g = g0 + tid;
    if (g > last / cpg || g >= G) return;
    const int lo = max(g * cpg, first) - first, hi = min((g + 1) * cpg, last + 1) - first;
    float s = 0.f, q = 0.f;
    for (int c = lo; c < hi; ++c) { s += chs1[c]; q += chs2[c]; }
    unsigned long long *dst = acc + ((long long)b * G + g) * 2;
    atomicAdd(dst, (unsigned long long)__float2ll_rn(s * L2D_GN_S1_SCALE));
    atomicAdd(dst + 1, (unsigned long long)__float2ll_rn(q * L2D_GN_S2_SCALE));
}

__device__ __forceinline__ float l2d_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif
