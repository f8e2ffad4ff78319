// Temporal attention kernels for gfx950 (HBM-bound, 1 flop/byte -> VALU + LDS, no MFMA).
//
// tattn_stream : the reference's StreamTemporalAttention core (stream_motion_module.py:99-213) in ONE pass
//   over the KV-cache:  scatter the new K/V row into cache slot update_idx[n] (pre-PE, in place),
//   K_l + k_pe[pe_idx[n,l]], q + q_pe[pe_idx[n,update_idx[n]]], 1xL scores + additive bias [N,L],
//   softmax, P.(V_l + v_pe[..]).  The reference makes >= 5 cache-sized passes (K+pe / V+pe temporaries,
//   head-reshape copies, .contiguous()); this kernel reads each cache byte exactly once.
// tattn_warmup : VersatileAttention (motion_module.py:469-530): bidirectional FxF attention over the
//   warm-up frames per pixel, writing the pre-PE K/V of the F frames into cache slots 0..F-1.
//
// Data layout / mapping.  Cache [N,2,T,L,C] fp16: for one (n, k|v, pixel) the L x C slab is contiguous.
// A thread owns one 8-channel (16-byte) column `cc` of one pixel for all L slots, so
//   * every wavefront load instruction covers whole contiguous 16 B x (C/8) rows -> fully coalesced,
//   * the P.V accumulation is thread-local (no cross-lane traffic),
//   * the only cross-thread step is the per-head score reduction over the d/8 threads of a head, done
//     through LDS (padded rows, conflict-free float4 writes; same-address broadcast reads).
// Block = PB pixels x TP threads (TP = C/8), PB chosen so that the block is 256..320 threads.
// Masked slots (bias == -inf) are never read: their softmax weight is exactly 0.
//
// Which kernel runs (op.i[5], 0 = auto): the SD widths with L <= 16 go to the LDS-DMA ring kernel in tattn_ring.hip
// (variant 13, 1.00 vs 1.33 ms per cfg-2 frame); this file keeps the kernels for everything else -- register-resident
// (1: other widths, L <= 16) and chunked (2 / 3 / 4: L = 24, 40) -- plus the builds that were used to analyse the
// register-resident kernel on the GPU and are reachable only by asking for them: 5 (V loaded after the softmax),
// 6 (half-size blocks), 7 (PE rows in LDS; slower), 8 / 9 (ablations: no cache loads / no score reduction; NOT
// attention) and 10 / 11 / 12 (streaming probes with the same access pattern; NOT attention).
#include "tattn.h"

__device__ __forceinline__ float dot8(h16x8 a, h16x8 b) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)b[e];
    return s;
}

// ABL (ablation builds for the on-GPU phase analysis only, selected by variants 8/9; results are NOT valid attention):
//   1 = no cache loads (every slot uses the new row), 2 = no cross-thread score reduction.
template <int TP, int PB, int L, bool PV = true, int ABL = 0>
__global__ __launch_bounds__(TP *PB) void tattn_stream_kernel(TAttnArgs a) {
    constexpr int HG = TP / 8;  // threads per head (= d/8)
    constexpr int LP = L + 4;   // padded LDS row (floats)
    constexpr bool PRELOAD_V = PV && (L <= 24);   // V rows in flight together with the K rows
    extern __shared__ __attribute__((aligned(16))) float sp[];  // [PB*TP][LP]

    const int tid = threadIdx.x;
    const int p = tid / TP, cc = tid - p * TP;
    const long long NT = (long long)a.N * a.T;
    const long long pix = (long long)blockIdx.x * PB + p;
    const bool valid = pix < NT;
    const int n = valid ? (int)(pix / a.T) : 0;
    const long long t = valid ? pix - (long long)n * a.T : 0;
    const int C = a.C;
    const int u = (int)a.update_idx[n];
    const long long *pei = a.pe_idx + (long long)n * L;
    const h16 *bi = a.bias + (long long)n * L;

    h16x8 q8 = l2d_zero8(), k8 = l2d_zero8(), v8 = l2d_zero8();
    h16 *kc = a.cache + (((long long)n * 2 + 0) * a.T + t) * L * C + cc * 8;
    h16 *vc = a.cache + (((long long)n * 2 + 1) * a.T + t) * L * C + cc * 8;
    if (valid) {
        const h16 *src = a.qkv + pix * 3 * C + cc * 8;
        q8 = l2d_ld8(src);
        k8 = l2d_ld8(src + C);
        v8 = l2d_ld8(src + 2 * C);
        q8 = q8 + l2d_ld8(a.q_pe + pei[u] * C + cc * 8);   // fp16 add: same rounding point as the reference (:139)
        l2d_st8(kc + (long long)u * C, k8);                 // cache stores pre-PE projections (:117-119)
        l2d_st8(vc + (long long)u * C, v8);
    }

    float bl[L];
    h16x8 vreg[PRELOAD_V ? L : 1];
    float part[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        bl[l] = (float)bi[l];
        const bool live = valid && bl[l] > -1e30f;
        h16x8 kk = k8;
        if (ABL != 1 && l != u) kk = live ? l2d_ld8(kc + (long long)l * C) : l2d_zero8();
        if (PRELOAD_V) {
            h16x8 vv = v8;
            if (ABL != 1 && l != u) vv = live ? l2d_ld8(vc + (long long)l * C) : l2d_zero8();
            vreg[l] = vv;
        }
        h16x8 pe = l2d_ld8(a.k_pe + pei[l] * C + cc * 8);
        kk = kk + pe;                                        // fp16 rounding of K+pe as in the reference (:140)
        part[l] = dot8(q8, kk);
    }
    float *row = sp + (long long)tid * LP;
#pragma unroll
    for (int l = 0; l < L; l += 4) *reinterpret_cast<f32x4 *>(row + l) = (f32x4){part[l], part[l + 1], part[l + 2], part[l + 3]};
    __syncthreads();
    float s[L];
#pragma unroll
    for (int l = 0; l < L; ++l) s[l] = 0.f;
    const int gs = p * TP + (cc / HG) * HG;
#pragma unroll 2
    for (int j = 0; j < (ABL == 2 ? 1 : HG); ++j) {
        const float *r = sp + (long long)(gs + j) * LP;
#pragma unroll
        for (int l = 0; l < L; l += 4) {
            f32x4 x = *reinterpret_cast<const f32x4 *>(r + l);
            s[l] += x[0]; s[l + 1] += x[1]; s[l + 2] += x[2]; s[l + 3] += x[3];
        }
    }
    const float scale = rsqrtf((float)(C / a.H));
    float mx = -3.0e38f;
#pragma unroll
    for (int l = 0; l < L; ++l) { s[l] = s[l] * scale + bl[l]; mx = fmaxf(mx, s[l]); }
    float den = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) { s[l] = __expf(s[l] - mx); den += s[l]; }
    const float inv = 1.0f / den;
    const h16 *vc_late = vc;
    if (!PRELOAD_V) asm volatile("" : "+v"(vc_late));   // keeps the V loads below the barrier (bounded registers)
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const bool live = valid && bl[l] > -1e30f;
        h16x8 vv;
        if (PRELOAD_V) {
            vv = vreg[l];
        } else {
            vv = v8;
            if (l != u) vv = live ? l2d_ld8(vc_late + (long long)l * C) : l2d_zero8();
        }
        vv = vv + l2d_ld8(a.v_pe + pei[l] * C + cc * 8);     // (:141)
        const float pl = s[l] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += pl * (float)vv[e];
    }
    if (valid) {
        h16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (h16)o[e];
        l2d_st8(a.out + pix * C + cc * 8, ov);
    }
}

// Variant with the gathered positional-encoding rows (k_pe / v_pe [pe_idx[n,l]]) staged ONCE per block in LDS
// instead of being re-fetched by every thread for every slot: halves the global-load instruction count and
// frees the 2 x 64 VGPRs those loads kept in flight, so K and V rows (2L x 16 B per thread) fit ~170 VGPRs and
// two 320-thread blocks share a CU (the loads of one overlap the reduction / store phase of the other).
// LDS: [PB*TP][L+4] fp32 partial scores | k_pe rows [L][C] | v_pe rows [L][C] halfs.
template <int TP, int PB, int L>
__global__ __launch_bounds__(TP *PB) void tattn_stream_lds_kernel(TAttnArgs a) {
    constexpr int HG = TP / 8;
    constexpr int LP = L + 4;
    constexpr int C = TP * 8;
    extern __shared__ __attribute__((aligned(16))) float sp[];
    h16 *pek = reinterpret_cast<h16 *>(sp + TP * PB * LP);
    h16 *pev = pek + L * C;

    const int tid = threadIdx.x;
    const int p = tid / TP, cc = tid - p * TP;
    const long long NT = (long long)a.N * a.T;
    const long long pix0 = (long long)blockIdx.x * PB;
    const long long pix = pix0 + p;
    const bool valid = pix < NT;
    const int n = valid ? (int)(pix / a.T) : 0;
    const int n_blk = (int)(pix0 / a.T);            // the PE rows staged in LDS are those of the block's first pixel
    const long long t = valid ? pix - (long long)n * a.T : 0;
    const int u = (int)a.update_idx[n];
    const long long *pei = a.pe_idx + (long long)n * L;
    const h16 *bi = a.bias + (long long)n * L;

    h16x8 q8 = l2d_zero8(), k8 = l2d_zero8(), v8 = l2d_zero8();
    h16 *kc = a.cache + (((long long)n * 2 + 0) * a.T + t) * L * C + cc * 8;
    h16 *vc = a.cache + (((long long)n * 2 + 1) * a.T + t) * L * C + cc * 8;
    if (valid) {
        const h16 *src = a.qkv + pix * 3 * C + cc * 8;
        q8 = l2d_ld8(src);
        k8 = l2d_ld8(src + C);
        v8 = l2d_ld8(src + 2 * C);
        q8 = q8 + l2d_ld8(a.q_pe + pei[u] * C + cc * 8);
        l2d_st8(kc + (long long)u * C, k8);
        l2d_st8(vc + (long long)u * C, v8);
    }
    // all K and V rows of this thread in flight (masked slots are never read: live-mask in one register)
    unsigned livemask = 0;
    h16x8 kreg[L], vreg[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const bool live = valid && (float)bi[l] > -1e30f;
        livemask |= live ? (1u << l) : 0u;
        kreg[l] = k8;
        vreg[l] = v8;
        if (l != u) {
            kreg[l] = live ? l2d_ld8(kc + (long long)l * C) : l2d_zero8();
            vreg[l] = live ? l2d_ld8(vc + (long long)l * C) : l2d_zero8();
        }
    }
    // stage the gathered PE rows (L2 resident, 2 x L x C halfs) while the cache rows are in flight
    {
        const long long *peb = a.pe_idx + (long long)n_blk * L;
        for (int idx = tid; idx < L * TP; idx += TP * PB) {
            const int l = idx / TP, c8 = idx - l * TP;
            const long long row = peb[l] * C + c8 * 8;
            l2d_st8(pek + l * C + c8 * 8, l2d_ld8(a.k_pe + row));
            l2d_st8(pev + l * C + c8 * 8, l2d_ld8(a.v_pe + row));
        }
    }
    __syncthreads();
    const bool pe_lds = (n == n_blk);               // a block straddling two denoise rows falls back to global PE
    float *row = sp + (long long)tid * LP;
#pragma unroll
    for (int l = 0; l < L; l += 4) {
        f32x4 pr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const h16x8 pe = pe_lds ? l2d_ld8(pek + (l + r) * C + cc * 8) : l2d_ld8(a.k_pe + pei[l + r] * C + cc * 8);
            pr[r] = dot8(q8, kreg[l + r] + pe);      // fp16 rounding of K+pe as in the reference (:140)
        }
        *reinterpret_cast<f32x4 *>(row + l) = pr;
    }
    __syncthreads();
    float s[L];
#pragma unroll
    for (int l = 0; l < L; ++l) s[l] = 0.f;
    const int gs = p * TP + (cc / HG) * HG;
#pragma unroll 2
    for (int j = 0; j < HG; ++j) {
        const float *r = sp + (long long)(gs + j) * LP;
#pragma unroll
        for (int l = 0; l < L; l += 4) {
            f32x4 x = *reinterpret_cast<const f32x4 *>(r + l);
            s[l] += x[0]; s[l + 1] += x[1]; s[l + 2] += x[2]; s[l + 3] += x[3];
        }
    }
    const float scale = rsqrtf((float)(C / a.H));
    float mx = -3.0e38f;
    const h16 *bi_late = bi;
    asm volatile("" : "+v"(bi_late));   // re-read the (L1-resident) bias row here instead of holding L floats since kernel entry
#pragma unroll
    for (int l = 0; l < L; ++l) {
        s[l] = s[l] * scale + (float)bi_late[l];
        mx = fmaxf(mx, s[l]);
    }
    float den = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) { s[l] = __expf(s[l] - mx); den += s[l]; }
    const float inv = 1.0f / den;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const h16x8 pe = pe_lds ? l2d_ld8(pev + l * C + cc * 8) : l2d_ld8(a.v_pe + pei[l] * C + cc * 8);
        const h16x8 vv = vreg[l] + pe;               // (:141)
        const float pl = s[l] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += pl * (float)vv[e];
    }
    if (valid) {
        h16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (h16)o[e];
        l2d_st8(a.out + pix * C + cc * 8, ov);
    }
}

// Chunked variant: bounded register footprint for long windows (L = 24, 40) -- slots are processed CH at a
// time (CH independent 16-byte loads in flight per thread), scores and probabilities live in the thread's
// LDS row instead of L-sized register arrays.
template <int TP, int PB, int L, int CH>
__global__ __launch_bounds__(TP *PB) void tattn_stream_chunked_kernel(TAttnArgs a) {
    constexpr int HG = TP / 8;
    constexpr int LP = L + 4;
    static_assert(L % CH == 0, "L must be a multiple of CH");
    extern __shared__ __attribute__((aligned(16))) float sp[];

    const int tid = threadIdx.x;
    const int p = tid / TP, cc = tid - p * TP;
    const long long NT = (long long)a.N * a.T;
    const long long pix = (long long)blockIdx.x * PB + p;
    const bool valid = pix < NT;
    const int n = valid ? (int)(pix / a.T) : 0;
    const long long t = valid ? pix - (long long)n * a.T : 0;
    const int C = a.C;
    const int u = (int)a.update_idx[n];
    const long long *pei = a.pe_idx + (long long)n * L;
    const h16 *bi = a.bias + (long long)n * L;

    h16x8 q8 = l2d_zero8(), k8 = l2d_zero8(), v8 = l2d_zero8();
    h16 *kc = a.cache + (((long long)n * 2 + 0) * a.T + t) * L * C + cc * 8;
    h16 *vc = a.cache + (((long long)n * 2 + 1) * a.T + t) * L * C + cc * 8;
    if (valid) {
        const h16 *src = a.qkv + pix * 3 * C + cc * 8;
        q8 = l2d_ld8(src);
        k8 = l2d_ld8(src + C);
        v8 = l2d_ld8(src + 2 * C);
        q8 = q8 + l2d_ld8(a.q_pe + pei[u] * C + cc * 8);
        l2d_st8(kc + (long long)u * C, k8);
        l2d_st8(vc + (long long)u * C, v8);
    }
    float *row = sp + (long long)tid * LP;
#pragma unroll 1
    for (int l0 = 0; l0 < L; l0 += CH) {
        h16x8 kk[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int l = l0 + i;
            const bool live = valid && (float)bi[l] > -1e30f;
            kk[i] = k8;
            if (l != u) kk[i] = live ? l2d_ld8(kc + (long long)l * C) : l2d_zero8();
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int l = l0 + i;
            h16x8 x = kk[i] + l2d_ld8(a.k_pe + pei[l] * C + cc * 8);
            row[l] = dot8(q8, x);
        }
    }
    __syncthreads();
    float s[L];
#pragma unroll
    for (int l = 0; l < L; ++l) s[l] = 0.f;
    const int gs = p * TP + (cc / HG) * HG;
#pragma unroll 2
    for (int j = 0; j < HG; ++j) {
        const float *r = sp + (long long)(gs + j) * LP;
#pragma unroll
        for (int l = 0; l < L; l += 4) {
            f32x4 x = *reinterpret_cast<const f32x4 *>(r + l);
            s[l] += x[0]; s[l + 1] += x[1]; s[l + 2] += x[2]; s[l + 3] += x[3];
        }
    }
    const float scale = rsqrtf((float)(C / a.H));
    float mx = -3.0e38f;
#pragma unroll
    for (int l = 0; l < L; ++l) { s[l] = s[l] * scale + (float)bi[l]; mx = fmaxf(mx, s[l]); }
    float den = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) { s[l] = __expf(s[l] - mx); den += s[l]; }
    const float inv = 1.0f / den;
    __syncthreads();   // every thread has finished reading the partial-score rows
#pragma unroll
    for (int l = 0; l < L; l += 4)
        *reinterpret_cast<f32x4 *>(row + l) = (f32x4){s[l] * inv, s[l + 1] * inv, s[l + 2] * inv, s[l + 3] * inv};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll 1
    for (int l0 = 0; l0 < L; l0 += CH) {
        h16x8 vv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int l = l0 + i;
            const bool live = valid && (float)bi[l] > -1e30f;
            vv[i] = v8;
            if (l != u) vv[i] = live ? l2d_ld8(vc + (long long)l * C) : l2d_zero8();
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int l = l0 + i;
            h16x8 x = vv[i] + l2d_ld8(a.v_pe + pei[l] * C + cc * 8);
            const float pl = row[l];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += pl * (float)x[e];
        }
    }
    if (valid) {
        h16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (h16)o[e];
        l2d_st8(a.out + pix * C + cc * 8, ov);
    }
}

// Streaming probes with the stream kernel's exact geometry (variants 10/11/12, analysis only -- the output is NOT
// attention): MODE 0 = sum of all K and V rows, 1 = K rows only, 2 = K + V rows + the gathered PE rows.  They
// bound what this access pattern can reach with no dependent index loads, no LDS exchange and no softmax.
#ifdef L2D_PROBES
template <int TP, int PB, int L, int MODE>
__global__ __launch_bounds__(TP *PB) void tattn_probe_kernel(TAttnArgs a) {
    const int tid = threadIdx.x;
    const int p = tid / TP, cc = tid - p * TP;
    const long long NT = (long long)a.N * a.T;
    const long long pix = (long long)blockIdx.x * PB + p;
    if (pix >= NT) return;
    const int n = (int)(pix / a.T);
    const long long t = pix - (long long)n * a.T;
    const int C = a.C;
    const h16 *kc = a.cache + (((long long)n * 2 + 0) * a.T + t) * L * C + cc * 8;
    const h16 *vc = a.cache + (((long long)n * 2 + 1) * a.T + t) * L * C + cc * 8;
    const long long *pei = a.pe_idx + (long long)n * L;
    h16x8 acc = l2d_zero8();
#pragma unroll
    for (int l = 0; l < L; ++l) {
        acc = acc + l2d_ld8(kc + (long long)l * C);
        if (MODE != 1) acc = acc + l2d_ld8(vc + (long long)l * C);
        if (MODE == 2) acc = acc + l2d_ld8(a.k_pe + pei[l] * C + cc * 8) + l2d_ld8(a.v_pe + pei[l] * C + cc * 8);
    }
    l2d_st8(a.out + pix * C + cc * 8, acc);
}
#endif

template <int TP, int PB, int L>
static int launch_stream_t(const TAttnArgs &a, hipStream_t s) {
    long long NT = (long long)a.N * a.T;
    int nb = (int)((NT + PB - 1) / PB);
    size_t lds = (size_t)TP * PB * (L + 4) * sizeof(float);
    // variant 0 (auto): register-resident for L <= 16, chunked (CH = 8) beyond; 1 / 2 / 3 force
    // register-resident / chunked CH=8 / chunked CH=4 (used by the on-GPU A/B in bench.py --tune)
    int v = a.variant;
    if (v == 0) v = (L <= 16) ? 1 : 2;
    if (v == 1 && L <= 16)
        hipLaunchKernelGGL((tattn_stream_kernel<TP, PB, (L <= 16 ? L : 16)>), dim3(nb), dim3(TP * PB), lds, s, a);
#ifdef L2D_PROBES   // analysis-only builds (`make PROBES=1`): ablations / streaming probes whose output is NOT attention
    else if (v == 8 && L <= 16)
        hipLaunchKernelGGL((tattn_stream_kernel<TP, PB, (L <= 16 ? L : 16), true, 1>), dim3(nb), dim3(TP * PB), lds, s, a);
    else if (v == 9 && L <= 16)
        hipLaunchKernelGGL((tattn_stream_kernel<TP, PB, (L <= 16 ? L : 16), true, 2>), dim3(nb), dim3(TP * PB), lds, s, a);
    else if (v >= 10 && v <= 12 && L <= 16) {
        constexpr int LL = (L <= 16 ? L : 16);
        if (v == 10) hipLaunchKernelGGL((tattn_probe_kernel<TP, PB, LL, 0>), dim3(nb), dim3(TP * PB), 0, s, a);
        if (v == 11) hipLaunchKernelGGL((tattn_probe_kernel<TP, PB, LL, 1>), dim3(nb), dim3(TP * PB), 0, s, a);
        if (v == 12) hipLaunchKernelGGL((tattn_probe_kernel<TP, PB, LL, 2>), dim3(nb), dim3(TP * PB), 0, s, a);
    }
#else
    else if (v >= 8 && v <= 12) {
        l2d_set_error("tattn_stream: variants 8-12 are analysis probes (output is not attention); build with `make PROBES=1`");
        return L2D_EINVAL;
    }
#endif
    else if (v == 7 && L <= 16)   // PE rows staged in LDS
        hipLaunchKernelGGL((tattn_stream_lds_kernel<TP, PB, (L <= 16 ? L : 16)>), dim3(nb), dim3(TP * PB),
                           lds + (size_t)2 * L * TP * 8 * sizeof(h16), s, a);
    else if (v == 5 && L <= 16)   // register resident, V loaded after the softmax: fewer VGPRs -> 2 blocks/CU
        hipLaunchKernelGGL((tattn_stream_kernel<TP, PB, (L <= 16 ? L : 16), false>), dim3(nb), dim3(TP * PB), lds, s, a);
    else if (v == 6 && L <= 16 && PB >= 2) {   // half the pixels per block: more, smaller blocks
        constexpr int PB2 = PB >= 2 ? PB / 2 : 1;
        int nb2 = (int)((NT + PB2 - 1) / PB2);
        hipLaunchKernelGGL((tattn_stream_kernel<TP, PB2, (L <= 16 ? L : 16)>), dim3(nb2), dim3(TP * PB2),
                           (size_t)TP * PB2 * (L + 4) * sizeof(float), s, a);
    }
    else if (v == 3)
        hipLaunchKernelGGL((tattn_stream_chunked_kernel<TP, PB, L, 4>), dim3(nb), dim3(TP * PB), lds, s, a);
    else if (v == 4 && L % 16 == 0)   // all K rows, then all V rows in flight (16 loads/thread), ~3 blocks/CU
        hipLaunchKernelGGL((tattn_stream_chunked_kernel<TP, PB, L, (L % 16 == 0 ? 16 : 4)>), dim3(nb), dim3(TP * PB), lds, s, a);
    else
        hipLaunchKernelGGL((tattn_stream_chunked_kernel<TP, PB, L, (L % 8 == 0 ? 8 : 4)>), dim3(nb), dim3(TP * PB), lds, s, a);
    return L2D_OK;
}

template <int TP, int PB>
static int launch_stream_l(const TAttnArgs &a, hipStream_t s) {
    switch (a.L) {
        case 12: return launch_stream_t<TP, PB, 12>(a, s);
        case 16: return launch_stream_t<TP, PB, 16>(a, s);
        case 24: return launch_stream_t<TP, PB, 24>(a, s);
        case 40: return launch_stream_t<TP, PB, 40>(a, s);
    }
    l2d_set_error("tattn_stream: unsupported window L=%d (built: 12,16,24,40)", a.L);
    return L2D_EINVAL;
}

static int tattn_common(const l2d_op *op, TAttnArgs &a, const char *what) {
    a.qkv = (const h16 *)op->p[0]; a.cache = (h16 *)op->p[1];
    a.q_pe = (const h16 *)op->p[2]; a.k_pe = (const h16 *)op->p[3]; a.v_pe = (const h16 *)op->p[4];
    a.pe_idx = (const long long *)op->p[5]; a.update_idx = (const long long *)op->p[6];
    a.bias = (const h16 *)op->p[7]; a.out = (h16 *)op->p[8];
    a.N = op->i[0]; a.T = op->i[1]; a.C = op->i[2]; a.L = op->i[3]; a.H = op->i[4]; a.variant = op->i[5];
#ifdef L2D_PROBES
    {
        extern unsigned long long *l2d_tattn_probe_ptr();
        a.probe = l2d_tattn_probe_ptr();
    }
#endif
    if (!a.qkv || !a.cache || !a.q_pe || !a.k_pe || !a.v_pe || !a.out || a.N <= 0 || a.T <= 0 || a.H != 8 ||
        (a.C % 64) || a.L <= 0) {
        l2d_set_error("%s(tag %d): invalid arguments (N=%d T=%d C=%d L=%d H=%d; need H=8, C%%64==0)", what, op->tag, a.N,
                      a.T, a.C, a.L, a.H);
        return L2D_EINVAL;
    }
    return L2D_OK;
}

int l2d_launch_tattn_stream(const l2d_op *op, hipStream_t s) {
    TAttnArgs a;
    int rc = tattn_common(op, a, "tattn_stream");
    if (rc) return rc;
    if (!a.pe_idx || !a.update_idx || !a.bias) {
        l2d_set_error("tattn_stream(tag %d): pe_idx/update_idx/bias missing", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    // LDS-DMA ring kernel (tattn_ring.hip): the default wherever it applies (1.00 vs 1.33 ms per cfg-2 frame)
    if (a.variant == 13 || (a.variant == 0 && l2d_tattn_ring_ok(a, op->p[9]))) {
        if (!l2d_tattn_ring_ok(a, op->p[9])) {
            l2d_set_error("tattn_stream(tag %d): ring variant needs C in {320,640,1280}, L in {12,16,24,40}, T %% 8 == 0, p9 = zero page", op->tag);
            return L2D_EINVAL;
        }
        rc = l2d_launch_tattn_ring(a, op->p[9], s);
        if (rc) return rc;
        return l2d_check_launch("tattn_stream_ring", op->tag);
    }
    switch (a.C) {
        case 64: rc = launch_stream_l<8, 32>(a, s); break;
        case 128: rc = launch_stream_l<16, 16>(a, s); break;
        case 256: rc = launch_stream_l<32, 8>(a, s); break;
        case 320: rc = launch_stream_l<40, 8>(a, s); break;
        case 640: rc = launch_stream_l<80, 4>(a, s); break;
        case 1280: rc = launch_stream_l<160, 2>(a, s); break;
        default:
            l2d_set_error("tattn_stream(tag %d): unsupported C=%d (built: 64,128,256,320,640,1280)", op->tag, a.C);
            return L2D_EINVAL;
    }
    if (rc) return rc;
    return l2d_check_launch("tattn_stream", op->tag);
}

// ------------------------------------------------------------------------------------------- warm-up
template <int TP, int PB, int F>
__global__ __launch_bounds__(TP *PB) void tattn_warmup_kernel(TAttnArgs a) {
    constexpr int HG = TP / 8;
    constexpr int LP = F * F + 4;
    extern __shared__ __attribute__((aligned(16))) float sp[];
    const int tid = threadIdx.x;
    const int p = tid / TP, cc = tid - p * TP;
    const long long t = (long long)blockIdx.x * PB + p;
    const bool valid = t < a.T;
    const int C = a.C, L = a.L;
    h16x8 q[F], k[F], v[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        q[f] = k[f] = v[f] = l2d_zero8();
        if (valid) {
            const h16 *src = a.qkv + ((long long)f * a.T + t) * 3 * C + cc * 8;
            q[f] = l2d_ld8(src);
            k[f] = l2d_ld8(src + C);
            v[f] = l2d_ld8(src + 2 * C);
            // cache_row [2,T,L,C]: slots 0..F-1 take the pre-PE projections (motion_module.py:492-493)
            l2d_st8(a.cache + ((0LL * a.T + t) * L + f) * C + cc * 8, k[f]);
            l2d_st8(a.cache + ((1LL * a.T + t) * L + f) * C + cc * 8, v[f]);
        }
        q[f] = q[f] + l2d_ld8(a.q_pe + (long long)f * C + cc * 8);
        k[f] = k[f] + l2d_ld8(a.k_pe + (long long)f * C + cc * 8);
        v[f] = v[f] + l2d_ld8(a.v_pe + (long long)f * C + cc * 8);
    }
    float *row = sp + (long long)tid * LP;
#pragma unroll
    for (int fq = 0; fq < F; ++fq)
#pragma unroll
        for (int fk = 0; fk < F; fk += 4)
            *reinterpret_cast<f32x4 *>(row + fq * F + fk) =
                (f32x4){dot8(q[fq], k[fk]), dot8(q[fq], k[fk + 1]), dot8(q[fq], k[fk + 2]), dot8(q[fq], k[fk + 3])};
    __syncthreads();
    const int gs = p * TP + (cc / HG) * HG;
    const float scale = rsqrtf((float)(C / a.H));
#pragma unroll
    for (int fq = 0; fq < F; ++fq) {
        float s[F];
#pragma unroll
        for (int fk = 0; fk < F; ++fk) s[fk] = 0.f;
        for (int j = 0; j < HG; ++j) {
            const float *r = sp + (long long)(gs + j) * LP + fq * F;
#pragma unroll
            for (int fk = 0; fk < F; fk += 4) {
                f32x4 x = *reinterpret_cast<const f32x4 *>(r + fk);
                s[fk] += x[0]; s[fk + 1] += x[1]; s[fk + 2] += x[2]; s[fk + 3] += x[3];
            }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int fk = 0; fk < F; ++fk) { s[fk] *= scale; mx = fmaxf(mx, s[fk]); }
        float den = 0.f;
#pragma unroll
        for (int fk = 0; fk < F; ++fk) { s[fk] = __expf(s[fk] - mx); den += s[fk]; }
        const float inv = 1.0f / den;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int fk = 0; fk < F; ++fk) {
            const float pl = s[fk] * inv;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += pl * (float)v[fk][e];
        }
        if (valid) {
            h16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (h16)o[e];
            l2d_st8(a.out + ((long long)fq * a.T + t) * C + cc * 8, ov);
        }
    }
}

template <int TP, int PB>
static int launch_warm_t(const TAttnArgs &a, int F, hipStream_t s) {
    int nb = (a.T + PB - 1) / PB;
    size_t lds = (size_t)TP * PB * (F * F + 4) * sizeof(float);
    if (F == 8) hipLaunchKernelGGL((tattn_warmup_kernel<TP, PB, 8>), dim3(nb), dim3(TP * PB), lds, s, a);
    else hipLaunchKernelGGL((tattn_warmup_kernel<TP, PB, 4>), dim3(nb), dim3(TP * PB), lds, s, a);
    return L2D_OK;
}

int l2d_launch_tattn_warmup(const l2d_op *op, hipStream_t s) {
    TAttnArgs a;
    int rc = tattn_common(op, a, "tattn_warmup");
    if (rc) return rc;
    int F = a.N;  // i0 carries the number of warm-up frames
    if ((F != 8 && F != 4) || F > a.L) {
        l2d_set_error("tattn_warmup(tag %d): F=%d unsupported (built: F in {4, 8}, F <= L)", op->tag, F);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    // a block stages F*F partial scores per thread in LDS (272 B/thread at F=8): blocks of <= 160 threads
    switch (a.C) {
        case 64: rc = launch_warm_t<8, 16>(a, F, s); break;
        case 128: rc = launch_warm_t<16, 8>(a, F, s); break;
        case 256: rc = launch_warm_t<32, 4>(a, F, s); break;
        case 320: rc = launch_warm_t<40, 4>(a, F, s); break;
        case 640: rc = launch_warm_t<80, 2>(a, F, s); break;
        case 1280: rc = launch_warm_t<160, 1>(a, F, s); break;
        default:
            l2d_set_error("tattn_warmup(tag %d): unsupported C=%d", op->tag, a.C);
            return L2D_EINVAL;
    }
    if (rc) return rc;
    return l2d_check_launch("tattn_warmup", op->tag);
}
