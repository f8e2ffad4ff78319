// Flash attention on MFMA for gfx950: spatial self-attention (T x T, T up to 9216) and text
// cross-attention (T x 77) of the SD-1.5 transformer blocks.  Replaces the SDPA call inside diffusers'
// `Attention` that the reference reaches from attention.py:243 (attn1) and :250-255 (attn2).
//
//   O[q][:] = softmax_k( Q[q].K[k] / sqrt(d) ) . V[k][:]        per (batch b, head h), no mask
//
// "Swapped" formulation so that the probabilities never leave registers:
//   S^T = K.Q^T      A operand = K tile rows (keys, from LDS), B operand = Q (held in registers)
//   O^T += V^T.P^T   A operand = V^T tile rows (head-dim, from LDS), B operand = P^T
// With v_mfma_f32_16x16x32_f16 the S^T accumulator puts, in lane (q = lane&15, g = lane>>4), the keys
// {16*ks + 4g + r}; that is exactly a valid K-slot assignment for the B operand of the second MFMA
// (any permutation of the contraction index is legal as long as A and B agree), so P^T is built
// lane-locally with 8 cvt and zero cross-lane traffic; the matching V^T A-fragment is two 8-byte LDS
// reads.  V arrives already transposed ([B][H*d][Tk]): the projection GEMM that produces V is simply
// issued with its operand roles swapped (see unet_hip.py), so no transpose pass exists anywhere.
// Softmax row statistics need only 2 cross-lane steps (xor 16, 32).
//
// Block = 4 waves x 32 query rows = 128 queries of one (b, h); K / V^T tiles of 64 keys are staged
// through LDS (register prefetch of tile t+1 during compute of tile t; double-buffered when it fits).
// LDS row strides are padded to the conflict-free residues found by simulating the gfx950 lane-group
// tables: K rows (ds_read_b128) stride = DQK+16 halfs, V^T rows (ds_read_b64) stride = 72 halfs.
#include "common.h"

struct FAArgs {
    const h16 *q, *k, *vt;
    h16 *out;
    int B, H, d, Tq, Tk, ldq, ldk, ldvt, ldo;
    long long sq, sk, svt, so;
};

template <int D>
struct FACfg {
    static constexpr int DQK = ((D + 31) / 32) * 32;  // contraction length of QK^T, zero padded
    static constexpr int KK = DQK / 32;
    static constexpr int D16 = (D + 15) / 16;          // 16-row blocks of O^T
    static constexpr int KROW = DQK + 16;              // halfs
    static constexpr int VROW = 72;                    // halfs (64 keys + pad)
    static constexpr int KCH = (64 * (DQK / 8)) / 256;            // K staging chunks per thread
    static constexpr int VTOT = D16 * 16 * 8;                     // V^T staging chunks per tile
    static constexpr int VCH = (VTOT + 255) / 256;
    static constexpr int TILE_HALFS = 64 * KROW + D16 * 16 * VROW;
    static constexpr int NBUF = (TILE_HALFS * 2 * 2 <= 65536) ? 2 : 1;
    static constexpr bool ONES = (D % 16) != 0;        // a spare padded row of V^T carries ones (see load_tiles)
};

template <int D>
__device__ __forceinline__ void flash_attn_body(const FAArgs &a) {
    using Cf = FACfg<D>;
    constexpr int DQK = Cf::DQK, KK = Cf::KK, D16 = Cf::D16, KROW = Cf::KROW, VROW = Cf::VROW;
    constexpr int KCH = Cf::KCH, VCH = Cf::VCH, NBUF = Cf::NBUF;
    __shared__ __attribute__((aligned(16))) h16 smem[NBUF * Cf::TILE_HALFS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const h16 *qp = a.q + (long long)b * a.sq + h * D;
    const h16 *kp = a.k + (long long)b * a.sk + h * D;
    const h16 *vp = a.vt + (long long)b * a.svt + (long long)h * D * a.ldvt;
    h16 *op = a.out + (long long)b * a.so + h * D;

    // Q fragments (B operand): lane (j=li, g=lg) holds Q[q0 + qs*16 + j][kk*32 + 8g .. +7]
    h16x8 qf[2][KK];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            int qr = q0 + qs * 16 + li, dc = kk * 32 + lg * 8;
            qf[qs][kk] = (qr < a.Tq && dc < D) ? l2d_ld8(qp + (long long)qr * a.ldq + dc) : l2d_zero8();
        }

    h16x8 kreg[KCH], vreg[VCH];
    auto load_tiles = [&](int key0) {
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            int id = tid + 256 * j;
            int r = id / (DQK / 8), c8 = id - r * (DQK / 8);
            int key = key0 + r;
            kreg[j] = (key < a.Tk && c8 * 8 < D) ? l2d_ld8(kp + (long long)key * a.ldk + c8 * 8) : l2d_zero8();
        }
#pragma unroll
        for (int j = 0; j < VCH; ++j) {
            int id = tid + 256 * j;
            int dd = id >> 3, c = id & 7;
            int key = key0 + c * 8;
            h16x8 v = l2d_zero8();
            if (Cf::ONES && dd == D) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (h16)1.0f;   // ones row: O^T[D][q] = sum_k p(q,k), the softmax denominator
            }
            if (id < Cf::VTOT && dd < D && key < a.Tk) {
                v = l2d_ld8(vp + (long long)dd * a.ldvt + key);
                if (key + 8 > a.Tk) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (key + e >= a.Tk) v[e] = (h16)0.0f;   // padding columns may hold anything
                }
            }
            vreg[j] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        h16 *Ks = smem + buf * Cf::TILE_HALFS;
        h16 *Vs = Ks + 64 * KROW;
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
            int id = tid + 256 * j;
            int r = id / (DQK / 8), c8 = id - r * (DQK / 8);
            l2d_st8(Ks + r * KROW + c8 * 8, kreg[j]);
        }
#pragma unroll
        for (int j = 0; j < VCH; ++j) {
            int id = tid + 256 * j;
            if (id < Cf::VTOT) l2d_st8(Vs + (id >> 3) * VROW + (id & 7) * 8, vreg[j]);
        }
    };

    f32x4 oacc[D16][2];
#pragma unroll
    for (int ds = 0; ds < D16; ++ds)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) oacc[ds][qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrow[2] = {-1.0e30f, -1.0e30f}, lrow[2] = {0.f, 0.f};
    const float c2e = rsqrtf((float)D) * 1.4426950408889634f;   // 1/sqrt(d) * log2(e)

    const int nt = (a.Tk + 63) / 64;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) load_tiles((kt + 1) * 64);
        const h16 *Ks = smem + cur * Cf::TILE_HALFS;
        const h16 *Vs = Ks + 64 * KROW;

        f32x4 sacc[4][2];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                h16x8 kf = l2d_ld8(Ks + (ks * 16 + li) * KROW + kk * 32 + lg * 8);
#pragma unroll
                for (int qs = 0; qs < 2; ++qs)   // first K step accumulates onto the inline constant 0: no zero-fill
                    sacc[ks][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qs][kk], kk == 0 ? zero4 : sacc[ks][qs], 0, 0, 0);
            }
        // Softmax in the exp2 domain on the RAW scores: p = exp2(s * c2e - mref * c2e) is one FMA + one v_exp_f32.
        // `mref` is a per-row reference, not the running maximum: it is only raised (and O, l rescaled) when some row
        // of the wave exceeds it by more than 2^8 -- p <= 256 is exact enough in fp16 and l, O are fp32 -- so after the
        // first tile the rescale branch (wave-uniform) is almost never taken.  Keys beyond Tk exist in the last tile only.
        if ((kt + 1) * 64 > a.Tk) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int qs = 0; qs < 2; ++qs)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((kt * 64 + ks * 16 + lg * 4 + r) >= a.Tk) sacc[ks][qs][r] = -3.0e38f;
        }
        h16x8 pf[2][2];   // [c2][qs]
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            float mx = fmaxf(fmaxf(sacc[0][qs][0], sacc[0][qs][1]), fmaxf(sacc[0][qs][2], sacc[0][qs][3]));
#pragma unroll
            for (int ks = 1; ks < 4; ++ks)
                mx = fmaxf(fmaxf(mx, fmaxf(sacc[ks][qs][0], sacc[ks][qs][1])), fmaxf(sacc[ks][qs][2], sacc[ks][qs][3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (__any((mx - mrow[qs]) * c2e > 8.0f)) {
                const float mnew = fmaxf(mrow[qs], mx);
                const float alpha = __builtin_amdgcn_exp2f((mrow[qs] - mnew) * c2e);
                mrow[qs] = mnew;
                lrow[qs] *= alpha;
#pragma unroll
                for (int ds = 0; ds < D16; ++ds) oacc[ds][qs] *= alpha;
            }
            const float mneg = -mrow[qs] * c2e;
            float psum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[ks][qs][r], c2e, mneg));
                    if (!Cf::ONES) psum += pv;
                    pf[ks >> 1][qs][(ks & 1) * 4 + r] = (h16)pv;
                }
            if (!Cf::ONES) lrow[qs] += psum;
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int ds = 0; ds < D16; ++ds) {
                const h16 *vr = Vs + (ds * 16 + li) * VROW + c2 * 32 + lg * 4;
                h16x4 lo = *reinterpret_cast<const h16x4 *>(vr);
                h16x4 hi = *reinterpret_cast<const h16x4 *>(vr + 16);
                h16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int qs = 0; qs < 2; ++qs)
                    oacc[ds][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[c2][qs], oacc[ds][qs], 0, 0, 0);
            }
        if (NBUF == 2) {
            if (kt + 1 < nt) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
            __syncthreads();
            if (kt + 1 < nt) store_tiles(0);
            __syncthreads();
        }
    }
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        float l;
        if (Cf::ONES) {
            // O^T row D (fragment D/16, lane group (D%16)/4, register D%4) holds the denominator of query li
            l = __shfl(oacc[D / 16][qs][D % 4], ((D % 16) / 4) * 16 + li, 64);
        } else {
            l = lrow[qs];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int qr = q0 + qs * 16 + li;
        if (qr >= a.Tq) continue;
#pragma unroll
        for (int ds = 0; ds < D16; ++ds) {
            int dc = ds * 16 + lg * 4;
            if (dc >= D) continue;
            h16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (h16)(oacc[ds][qs][r] * inv);
            *reinterpret_cast<h16x4 *>(op + (long long)qr * a.ldo + dc) = o;
        }
    }
}

// Two entry points over the same body: with the occupancy hint the compiler keeps the accumulators in plain VGPRs (no
// AGPR <-> VGPR copies around every softmax) and fits d = 40 in 162 and d = 80 in 200 registers (188 / 264 without:
// d = 80 ran ONE wave per SIMD); d = 160 would spill under the hint and keeps the default budget.
template <int D>
__global__ __launch_bounds__(256) void flash_attn_kernel(FAArgs a) { flash_attn_body<D>(a); }
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void flash_attn_kernel_w2(FAArgs a) { flash_attn_body<D>(a); }

template <int D>
static void launch_fa(const FAArgs &a, hipStream_t s) {
    dim3 grid((a.Tq + 127) / 128, a.H, a.B);
    if (D <= 80) hipLaunchKernelGGL((flash_attn_kernel_w2<D>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((flash_attn_kernel<D>), grid, dim3(256), 0, s, a);
}

int l2d_launch_flash_ring(const l2d_op *op, int qs, hipStream_t s);   // flash_attn_ring.hip

int l2d_launch_flash_attn(const l2d_op *op, hipStream_t s) {
    FAArgs a;
    a.q = (const h16 *)op->p[0]; a.k = (const h16 *)op->p[1]; a.vt = (const h16 *)op->p[2]; a.out = (h16 *)op->p[3];
    a.B = op->i[0]; a.H = op->i[1]; a.d = op->i[2]; a.Tq = op->i[3]; a.Tk = op->i[4];
    a.ldq = op->i[5]; a.ldk = op->i[6]; a.ldvt = op->i[7]; a.ldo = op->i[8];
    a.sq = op->l[0]; a.sk = op->l[1]; a.svt = op->l[2]; a.so = op->l[3];
    if (!a.q || !a.k || !a.vt || !a.out || a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tk <= 0 || (a.ldq % 8) || (a.ldk % 8) ||
        (a.ldvt % 8) || (a.ldo % 4) || a.ldvt < ((a.Tk + 7) / 8) * 8) {
        l2d_set_error("flash_attn(tag %d): invalid arguments (B=%d H=%d d=%d Tq=%d Tk=%d ldvt=%d)", op->tag, a.B, a.H, a.d,
                      a.Tq, a.Tk, a.ldvt);
        return L2D_EINVAL;
    }
    const int variant = op->i[9];   // 0 auto (LDS-DMA ring kernel when its alignment rules hold), 1 register-staged kernel,
                                    // 2 / 3 ring kernel with 32 / 16 query rows per wave, 4 ring kernel with the pipelined loop
    if (variant < 0 || variant > 4) {
        l2d_set_error("flash_attn(tag %d): unknown variant %d", op->tag, variant);
        return L2D_EINVAL;
    }
    const bool ring_ok = op->p[4] && (a.d % 8) == 0 && ((((uintptr_t)a.k) | ((uintptr_t)a.vt)) & 15) == 0 &&
                         ((a.sk | a.svt) % 8) == 0;
    if (variant >= 2 && !ring_ok) {
        l2d_set_error("flash_attn(tag %d): ring variant needs p4 = zero page, d %% 8 == 0 and 16-byte aligned K / V^T", op->tag);
        return L2D_EINVAL;
    }
    L2D_DRY_RETURN();
    if (variant != 1 && ring_ok) {
        int rc = l2d_launch_flash_ring(op, variant >= 2 ? variant : 0, s);
        if (rc != L2D_OK) {
            l2d_set_error("flash_attn(tag %d): unsupported head dim %d (built: 8,16,32,40,64,80,160)", op->tag, a.d);
            return rc;
        }
        return l2d_check_launch("flash_attn_ring", op->tag);
    }
    switch (a.d) {
        case 8: launch_fa<8>(a, s); break;
        case 16: launch_fa<16>(a, s); break;
        case 32: launch_fa<32>(a, s); break;
        case 40: launch_fa<40>(a, s); break;
        case 64: launch_fa<64>(a, s); break;
        case 80: launch_fa<80>(a, s); break;
        case 160: launch_fa<160>(a, s); break;
        default:
            l2d_set_error("flash_attn(tag %d): unsupported head dim %d (built: 8,16,32,40,64,80,160)", op->tag, a.d);
            return L2D_EINVAL;
    }
    return l2d_check_launch("flash_attn", op->tag);
}
