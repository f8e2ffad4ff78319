// Streaming temporal attention, LDS-DMA ring kernel (SD widths C = 320 / 640 / 1280, window L = 12 / 16 / 24 / 40).
// Same math and rounding points as tattn_stream_kernel (tattn.hip; reference stream_motion_module.py:99-213).
//
// Why: the register-resident kernel issues all 2L loads of a pixel, waits, then computes; per cfg-2 frame that is
// 0.63 ms of streaming (probe, 4.85 TB/s) PLUS 0.49 ms of compute = 1.33 ms measured -- nothing overlaps, because a
// launch is only ~2 rounds of blocks and co-resident blocks run in lockstep.  Here the K / V slabs stream
// HBM -> LDS through a ring of `global_load_lds` stages (no VGPRs held by loads in flight), so the rows of the next
// stages / the next pixel group land while the current stage's dot products, the score exchange and the softmax
// run.
//
// Mapping.  Work item = (pixel t, 320-channel chunk): TP = 40 threads x 8 channels, so every width has the same
// geometry.  A block owns ONE (batch row n, chunk) and `gpb` consecutive groups of PB = 8 pixels (320 threads:
// thread = (p, cc)), so update_idx / bias / pe_idx and the gathered positional-encoding rows (2 x L x 640 B, staged
// once in LDS) are block constants.  A stage is R = 4 cache rows of the group's K (then V): 4 rows x 8 pixels x
// 640 B = 20 KB, laid out [r][p][cc] = [r][tid]: DMA instruction j of a stage moves row l0 + j for all 320 threads,
// thread tid fetching the 16 bytes it will consume itself.  So the row, its source (cache / new row / masked) and
// the source base address are wave-uniform (SALU), the per-thread part of the address is one constant 32-bit
// offset, and both the DMA's LDS image and the consumer's ds_read_b128 are lane-linear (conflict free).
// Per group: L/4 K stages -> score exchange over the d/8 threads of a head + softmax -> L/4 V stages.
// Masked slots (bias = -inf) are never fetched from the cache: their DMA reads the zero page (every thread still
// issues exactly LPS loads per stage, which keeps the counted vmcnt waits valid) and their softmax weight is
// exactly 0.  The new row (slot update_idx[n]) is DMA'd from the qkv projection buffer instead of the cache and
// written to the cache from LDS by its consumer.
//
// VMEM discipline: everything the inner loop needs comes through the ring or LDS.  A register spill or an
// ordinary global load whose value is needed soon would sit in the same in-order VMEM queue BEHIND the ring's
// outstanding stages and drain the pipeline, so the only ordinary loads are the next group's q row (issued L/8
// stages before use) and the stores (never waited on).  The loop is instruction-issue bound on the SIMD that hosts
// two of the block's five waves, hence v_dot2_f32_f16 / v_fma_mix_f32 and uniform control flow throughout.
#include <stdlib.h>

#include "tattn.h"

#define L2D_GPTR(p) ((__attribute__((address_space(1))) const void *)(p))
#define L2D_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef L2D_PROBES
static unsigned long long *g_tattn_probe = nullptr;
extern "C" void l2d_tattn_set_probe(void *p) { g_tattn_probe = (unsigned long long *)p; }
unsigned long long *l2d_tattn_probe_ptr() { return g_tattn_probe; }
// stamps of the second pixel group's stage 1 (a K stage) and stage NSTG/2 + 1 (a V stage): [block][wave][2][8]
#define TR_STAMP(gi, s, i)                                                                                          \
    do {                                                                                                            \
        if (a.probe && (gi) == 1 && ((s) == 1 || (s) == NSTG / 2 + 1) && (threadIdx.x & 63) == 0)                    \
            a.probe[(((unsigned long long)blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + ((s) == 1 ? 0 : 1)) * 8 + (i)] = \
                __builtin_readcyclecounter();                                                                       \
    } while (0)
#else
#define TR_STAMP(gi, s, i) do { } while (0)
#endif

// dot product on v_dot2_f32_f16: two exact fp16 products + fp32 accumulate per instruction
__device__ __forceinline__ float ring_dot8(h16x8 a, h16x8 b) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e += 2) s = __builtin_amdgcn_fdot2((h16x2){a[e], a[e + 1]}, (h16x2){b[e], b[e + 1]}, s, false);
    return s;
}

// o[0..7] += p * float(v[0..7]) on v_fma_mix_f32 (fp16 operand converted inside the FMA: no separate v_cvt)
__device__ __forceinline__ void ring_axpy8(float (&o)[8], float p, h16x8 v) {
    const u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(o[2 * e]) : "v"(w[e]), "v"(p));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(o[2 * e + 1]) : "v"(w[e]), "v"(p));
    }
}

#ifdef L2D_PROBES
// Round-2 form: ANALYSIS BUILDS ONLY (make PROBES=1; L2D_TATTN_RING=4 selects it there) -- the A/B partner of the loader-wave
// kernel below and the carrier of the in-kernel stage stamps (tools/tattn_probe.py); the product library does not contain it
// (round 4).  Every one of the five waves issues its own share of a stage's refill and then
// does the stage's arithmetic.  HG = threads per head (d / 8): 5, 10, 20.  R cache rows per ring stage, NS stages: (4, 5) =
// 100 KB of ring, one block of PB = 8 pixels x 40 threads per CU.  Geometries that were measured and removed again (round 2 /
// round 3, profiles/r2l*, round3_e_*): (2 rows, 3 stages) with two blocks per CU (10 % slower), 16-pixel / 10-wave blocks (equal),
// refill DMAs interleaved row by row with the arithmetic (2.7 % slower).
template <int HG, int L, int R, int NS, int PB>
__global__ __launch_bounds__(40 * PB) void tattn_stream_ring_kernel(TAttnArgs a, const h16 *zero, int gpb, int groups_per_unit) {
    constexpr int TP = 40, NT = TP * PB, NSTG = 2 * L / R, LPS = R;
    constexpr int STAGE_H = R * PB * TP * 8;       // halfs per stage (20 KB at R = 4, PB = 8)
    constexpr int LP = L + 4;
    // LDS: ring [NS][stage] | score rows [NT][LP] f32 | bias of row n [L] f32 | k_pe rows [L][40][8] | v_pe rows
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_raw[];
    h16 *ring = reinterpret_cast<h16 *>(ring_raw);
    float *sp = reinterpret_cast<float *>(ring_raw + (size_t)NS * STAGE_H * sizeof(h16));
    float *blds = sp + NT * LP;
    h16 *kpl = reinterpret_cast<h16 *>(blds + L);
    h16 *vpl = kpl + L * 320;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = tid / TP, cc = tid - p * TP;
    const int C = a.C, CH = C / 320;
    // block -> (n, chunk, first group); the divisions run on the VALU, pin the results to SGPRs
    const int bpu = (groups_per_unit + gpb - 1) / gpb;
    const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x / bpu);
    const int n = __builtin_amdgcn_readfirstlane(unit / CH);
    const int chunk = unit - n * CH;
    const int g0 = (blockIdx.x - unit * bpu) * gpb;
    const int ng = min(gpb, groups_per_unit - g0);
    const long long slab = (long long)a.T * L * C;
    // uniform bases (SGPRs) of the block's first pixel; per-thread parts are the 32-bit offsets below
    h16 *kbase = a.cache + (long long)n * 2 * slab + (long long)g0 * PB * L * C + chunk * 320;
    h16 *vbase = kbase + slab;
    const h16 *qkv_b = a.qkv + ((long long)n * a.T + g0 * PB) * 3 * C + chunk * 320;
    h16 *out_b = a.out + ((long long)n * a.T + g0 * PB) * C + chunk * 320;
    const unsigned coff = (unsigned)(p * L * C + cc * 8);       // cache: + l * C, + group * PB*L*C
    const unsigned qoff = (unsigned)(p * 3 * C + cc * 8);       // qkv:   + C (k) / 2C (v), + group * PB*3C
    const unsigned ooff = (unsigned)(p * C + cc * 8);           // out:   + group * PB*C

    // ---- block constants of batch row n
    const int u = __builtin_amdgcn_readfirstlane((int)a.update_idx[n]);
    const long long *pei = a.pe_idx + (long long)n * L;
    const h16 *bi = a.bias + (long long)n * L;
    unsigned long long live = 0;                    // bit l: slot l is fetched from the cache (L up to 40: 64 bits)
#pragma unroll
    for (int l = 0; l < L; ++l)
        if ((float)bi[l] > -1e30f && l != u) live |= 1ull << l;
    live = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(live >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(live & 0xffffffffull));
    if (tid < L) blds[tid] = (float)bi[tid];        // visible after the first stage barrier
    // gathered PE rows of this chunk -> LDS by DMA (older than every ring load: landed before the first stage is
    // consumed, visible to the block after that stage's barrier).  L * 40 items of 16 B per table.
#pragma unroll
    for (int f0 = 0; f0 < L * TP; f0 += NT) {
        const int f = f0 + tid;
        if (f < L * TP) {
            const int l = f / TP, c = f - l * TP;
            const long long po = pei[l] * C + chunk * 320 + c * 8;
            __builtin_amdgcn_global_load_lds(L2D_GPTR(a.k_pe + po), L2D_LPTR(kpl + (f0 + wave * 64) * 8), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(L2D_GPTR(a.v_pe + po), L2D_LPTR(vpl + (f0 + wave * 64) * 8), 16, 0, 0);
        }
    }
    const h16x8 qpe = l2d_ld8(a.q_pe + pei[u] * C + chunk * 320 + cc * 8);

    // ---- DMA side
    const int total = ng * NSTG;
    int it_issue = 0, is_s = 0, is_slot = 0, is_g = 0;
    // one refill = LPS row DMAs + the bookkeeping.  A `global_load_lds` wave-instruction takes ~200 cycles to issue in this
    // kernel (profiles/r3r_tattn_stage_phases.txt): four of them in front of the arithmetic make a stage the SUM of refill issue
    // and arithmetic -- which is what the loader-wave form below removes
    auto issue_row = [&](int j) {
        const bool isv = is_s >= NSTG / 2;
        const int l0 = (isv ? is_s - NSTG / 2 : is_s) * R;
        const h16 *cb = (isv ? vbase : kbase) + (long long)is_g * (PB * L * C);
        const h16 *qb = qkv_b + (long long)is_g * (PB * 3 * C) + (isv ? 2 * C : C);
        h16 *dst = ring + is_slot * STAGE_H + wave * 512;
        // Source of row l: the cache (live slot), the new projection row (slot u) or the zero page (masked slot).  All three
        // conditions are wave-uniform, but written as if / else the compiler emits three taken-or-not branches and re-materialised
        // 64-bit pointers per row (~28 instructions between two DMAs).  Branch-free instead: masks on the scalar unit select the
        // 64-bit base, two VALU ANDs select the per-thread offset.
        const int l = l0 + j;                                // uniform
        const unsigned long long lv = 0ull - ((live >> l) & 1ull);                          // all ones: live slot
        const unsigned long long nw = (0ull - (unsigned long long)(l == u ? 1u : 0u)) & ~lv; // all ones: the new row
        const unsigned long long sb = ((unsigned long long)(cb + l * C) & lv) | ((unsigned long long)qb & nw) |
                                      ((unsigned long long)zero & ~(lv | nw));
        const unsigned vo = (coff & (unsigned)lv) | (qoff & (unsigned)nw);
        const h16 *src = reinterpret_cast<const h16 *>(sb) + vo;
        __builtin_amdgcn_global_load_lds(L2D_GPTR(src), L2D_LPTR(dst + j * NT * 8), 16, 0, 0);
    };
    auto issue_done = [&]() {
        ++it_issue;
        if (++is_s == NSTG) { is_s = 0; ++is_g; }
        is_slot = (is_slot + 1 == NS) ? 0 : is_slot + 1;
    };
    auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < LPS; ++j) issue_row(j);
        issue_done();
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (it_issue < total) issue();

    const float scale = rsqrtf((float)(C / a.H));
    const int gs = p * TP + (cc / HG) * HG;
    float *row = sp + tid * LP;
    const h16 *kpt = kpl + cc * 8, *vpt = vpl + cc * 8;           // + l * 320
    int it = 0, cs_slot = 0;
    h16x8 qn = l2d_zero8();
    for (int gi = 0; gi < ng; ++gi) {
        h16x8 q8 = qn;
        if (gi == 0) q8 = l2d_ld8(qkv_b + qoff);
        q8 = q8 + qpe;                                           // fp16 add, as the reference (:139)
        h16 *ku = kbase + (long long)gi * (PB * L * C) + u * C;  // uniform: slot u of the group's first pixel
        float sc[L];
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int s = 0; s < NSTG; ++s) {
            // stage `it` has landed when at most NS-2 younger stages are outstanding (in-order return; other VMEM ops
            // in flight only make this wait longer, never shorter)
            TR_STAMP(gi, s, 0);                                  // stage top
            if (total - 1 - it >= NS - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            TR_STAMP(gi, s, 1);                                  // own DMA share of this stage landed
            __builtin_amdgcn_s_barrier();
            TR_STAMP(gi, s, 2);                                  // barrier passed
            const bool refill = it_issue < total;                // refills the slot every wave finished one stage ago
            if (refill) issue();
            TR_STAMP(gi, s, 3);                                  // refill issued 
            const h16 *st = ring + cs_slot * STAGE_H + tid * 8;
            if (s < NSTG / 2) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int l = s * R + r;
                    h16x8 kk = l2d_ld8(st + r * NT * 8);         // masked slots hold zeros
                    if (l == u) l2d_st8(ku + coff, kk);          // cache keeps the pre-PE projections (:117-119)
                    kk = kk + l2d_ld8(kpt + l * 320);            // fp16 rounding of K+pe as in the reference (:140)
                    sc[l] = ring_dot8(q8, kk);
                }
                if (s == NSTG / 2 - 1) {
                    // per-head score reduction over the HG threads of a head, then the 1 x L softmax
#pragma unroll
                    for (int l = 0; l < L; l += 4) *reinterpret_cast<f32x4 *>(row + l) = (f32x4){sc[l], sc[l + 1], sc[l + 2], sc[l + 3]};
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
#pragma unroll
                    for (int l = 0; l < L; ++l) sc[l] = 0.f;
#pragma unroll 2
                    for (int j = 0; j < HG; ++j) {
                        const float *rr = sp + (gs + j) * LP;
#pragma unroll
                        for (int l = 0; l < L; l += 4) {
                            f32x4 x = *reinterpret_cast<const f32x4 *>(rr + l);
                            sc[l] += x[0]; sc[l + 1] += x[1]; sc[l + 2] += x[2]; sc[l + 3] += x[3];
                        }
                    }
                    float mx = -3.0e38f;
#pragma unroll
                    for (int l = 0; l < L; l += 4) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(blds + l);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { sc[l + e] = sc[l + e] * scale + b4[e]; mx = fmaxf(mx, sc[l + e]); }
                    }
                    float den = 0.f;
#pragma unroll
                    for (int l = 0; l < L; ++l) { sc[l] = __expf(sc[l] - mx); den += sc[l]; }
                    const float inv = 1.0f / den;
#pragma unroll
                    for (int l = 0; l < L; ++l) sc[l] *= inv;
                }
            } else {
                if (s == NSTG / 2 && gi + 1 < ng)                // next group's q row: in flight for NSTG/2 stages
                    qn = l2d_ld8(qkv_b + (long long)(gi + 1) * (PB * 3 * C) + qoff);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int l = (s - NSTG / 2) * R + r;
                    h16x8 vv = l2d_ld8(st + r * NT * 8);
                    if (l == u) l2d_st8(ku + slab + coff, vv);
                    vv = vv + l2d_ld8(vpt + l * 320);            // (:141)
                    ring_axpy8(o, sc[l], vv);
                }
            }
            TR_STAMP(gi, s, 4);                                  // stage arithmetic issued
            ++it;
            cs_slot = (cs_slot + 1 == NS) ? 0 : cs_slot + 1;
        }
        h16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (h16)o[e];
        l2d_st8(out_b + (long long)gi * (PB * C) + ooff, ov);
    }
}
#endif

// Loader-wave form (round 3, the default).  In the kernel above every wave issues its share of a stage's refill (4
// `global_load_lds`, ~200 cycles of issue EACH in this kernel) and then does the stage's arithmetic: a stage is the sum of
// both, and neither interleaving the two (measured: 2.7 % slower) nor more waves changes that, because issuing a DMA blocks the
// issuing wave.  Here a SIXTH wave does nothing but issue: it owns the DMA descriptors of all 320 work items (5
// wave-instructions per cache row), waits for its own loads with the counted vmcnt and meets the five consumer waves at the
// stage barrier; the consumers run the arithmetic and never touch the ring's VMEM queue.  Same LDS image, same stage order,
// same rounding points, same results bit for bit.  cfg-2 frame: 1.02 -> 0.89 ms (profiles/round3_f_tattn_loader_wave_ab.txt); two
// and four loader waves measured the same as one (round3_g): the loader is not what bounds a stage any more.
//
// Scores in LDS.  The first loader-wave kernel kept a row's L scores in registers across a fully unrolled stage loop (2 L / R
// stages).  At L = 40 that loop is 40 stages long, hipcc stops unrolling it and the score array lands in scratch (176 bytes
// per lane; 2.0 TB/s, no better than the chunked kernel of tattn.hip).  Here the stage loops are ordinary loops: the partial
// scores of a row go straight into the thread's LDS score row, the L-wide softmax runs once on registers between the K and
// the V stages, and the probabilities are written back to the thread's own score row, from where the V stages read them --
// two block barriers per pixel group for the exchange instead of one, 110 VGPRs instead of 154-221, no unrolled code.
// Measured against the unrolled form with the same ring geometry (profiles/round3_j_tattn_lds_scores_ab.txt): L = 16 0.895 ->
// 0.876 ms per cfg-2 frame, L = 24 (cfg-3) 1.89 -> 1.76 ms, and L = 40 (cfg-5, 17 GB of cache per frame) 8.82 -> 3.99 ms
// against the chunked kernel (0.25 -> 0.56 of 8 TB/s, profiles/round3_i_tattn_l40_ring_vs_chunked.txt) -- so this is the one form
// for every window.  Ring geometry: (4 rows, 5 stages) = 100 KB for L <= 16, (4, 4) for L = 24, (2, 5) for L = 40, where the
// gathered PE rows (2 x 25 KB) and the score rows (55 KB) leave 50 KB for the ring.
template <int HG, int L, int R, int NS>
__global__ __launch_bounds__(384) void tattn_stream_ringlw_kernel(TAttnArgs a, const h16 *zero, int gpb, int groups_per_unit) {
    constexpr int PB = 8, TP = 40, NT = TP * PB, NSTG = 2 * L / R, LPSL = 5 * R;
    constexpr int STAGE_H = R * PB * TP * 8;
    constexpr int LP = L + 4;
    static_assert((NS - 2) * LPSL < 64 && L % 4 == 0 && L % R == 0, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_raw[];
    h16 *ring = reinterpret_cast<h16 *>(ring_raw);
    float *sp = reinterpret_cast<float *>(ring_raw + (size_t)NS * STAGE_H * sizeof(h16));
    float *blds = sp + NT * LP;
    h16 *kpl = reinterpret_cast<h16 *>(blds + L);
    h16 *vpl = kpl + L * 320;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, CH = C / 320;
    const int bpu = (groups_per_unit + gpb - 1) / gpb;
    const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x / bpu);
    const int n = __builtin_amdgcn_readfirstlane(unit / CH);
    const int chunk = unit - n * CH;
    const int g0 = (blockIdx.x - unit * bpu) * gpb;
    const int ng = min(gpb, groups_per_unit - g0);
    const long long slab = (long long)a.T * L * C;
    h16 *kbase = a.cache + (long long)n * 2 * slab + (long long)g0 * PB * L * C + chunk * 320;
    h16 *vbase = kbase + slab;
    const h16 *qkv_b = a.qkv + ((long long)n * a.T + g0 * PB) * 3 * C + chunk * 320;
    h16 *out_b = a.out + ((long long)n * a.T + g0 * PB) * C + chunk * 320;
    const int u = __builtin_amdgcn_readfirstlane((int)a.update_idx[n]);
    const long long *pei = a.pe_idx + (long long)n * L;
    const h16 *bi = a.bias + (long long)n * L;
    const int total = ng * NSTG;

    if (wave == 5) {
        // ---------------------------------------------------------------- the loader wave
        const int lane = tid & 63;
        unsigned long long live = 0;
        for (int l = 0; l < L; ++l)
            if ((float)bi[l] > -1e30f && l != u) live |= 1ull << l;
        live = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(live >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(live & 0xffffffffull));
        unsigned coff[5], qoff[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int v = k * 64 + lane, p = v / TP, cc = v - p * TP;
            coff[k] = (unsigned)(p * L * C + cc * 8);
            qoff[k] = (unsigned)(p * 3 * C + cc * 8);
        }
        int it_issue = 0, is_s = 0, is_slot = 0, is_g = 0;
        auto issue = [&]() {
            const bool isv = is_s >= NSTG / 2;
            const int l0 = (isv ? is_s - NSTG / 2 : is_s) * R;
            const h16 *cb = (isv ? vbase : kbase) + (long long)is_g * (PB * L * C);
            const h16 *qb = qkv_b + (long long)is_g * (PB * 3 * C) + (isv ? 2 * C : C);
            h16 *dst = ring + is_slot * STAGE_H;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int l = l0 + j;
                const unsigned long long lv = 0ull - ((live >> l) & 1ull);
                const unsigned long long nw = (0ull - (unsigned long long)(l == u ? 1u : 0u)) & ~lv;
                const unsigned long long sb = ((unsigned long long)(cb + l * C) & lv) | ((unsigned long long)qb & nw) |
                                              ((unsigned long long)zero & ~(lv | nw));
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const unsigned vo = (coff[k] & (unsigned)lv) | (qoff[k] & (unsigned)nw);
                    __builtin_amdgcn_global_load_lds(L2D_GPTR(reinterpret_cast<const h16 *>(sb) + vo), L2D_LPTR(dst + j * NT * 8 + k * 512), 16, 0, 0);
                }
            }
            ++it_issue;
            if (++is_s == NSTG) { is_s = 0; ++is_g; }
            is_slot = (is_slot + 1 == NS) ? 0 : is_slot + 1;
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (it_issue < total) issue();
        int it = 0;
        for (int gi = 0; gi < ng; ++gi) {
#pragma unroll 1
            for (int s = 0; s < NSTG; ++s) {
                TR_STAMP(gi, s, 0);                          // loader: stage top
                if (total - 1 - it >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPSL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TR_STAMP(gi, s, 1);                          // stage `it` has landed
                __builtin_amdgcn_s_barrier();
                TR_STAMP(gi, s, 2);                          // barrier passed (the consumers finished stage it - 1)
                if (it_issue < total) issue();
                TR_STAMP(gi, s, 3);                          // refill issued
                TR_STAMP(gi, s, 4);
                if (s == NSTG / 2 - 1) {                     // the consumers' two softmax barriers
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_barrier();
                }
                ++it;
            }
        }
        return;
    }

    // ---------------------------------------------------------------- consumers
    const int p = tid / TP, cc = tid - p * TP;
    const unsigned coff = (unsigned)(p * L * C + cc * 8);
    const unsigned qoff = (unsigned)(p * 3 * C + cc * 8);
    const unsigned ooff = (unsigned)(p * C + cc * 8);
    if (tid < L) blds[tid] = (float)bi[tid];
    for (int f0 = 0; f0 < L * TP; f0 += NT) {
        const int f = f0 + tid;
        if (f < L * TP) {
            const int l = f / TP, c = f - l * TP;
            const long long po = pei[l] * C + chunk * 320 + c * 8;
            __builtin_amdgcn_global_load_lds(L2D_GPTR(a.k_pe + po), L2D_LPTR(kpl + (f0 + wave * 64) * 8), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(L2D_GPTR(a.v_pe + po), L2D_LPTR(vpl + (f0 + wave * 64) * 8), 16, 0, 0);
        }
    }
    const h16x8 qpe = l2d_ld8(a.q_pe + pei[u] * C + chunk * 320 + cc * 8);
    h16x8 qn = l2d_ld8(qkv_b + qoff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's PE-row DMAs have landed before the first barrier publishes them

    const float scale = rsqrtf((float)(C / a.H));
    const int gs = p * TP + (cc / HG) * HG;
    float *row = sp + tid * LP;
    const h16 *kpt = kpl + cc * 8, *vpt = vpl + cc * 8;
    int cs_slot = 0;
    for (int gi = 0; gi < ng; ++gi) {
        const h16x8 q8 = qn + qpe;
        h16 *ku = kbase + (long long)gi * (PB * L * C) + u * C;
        // ---- K stages: partial scores of this thread's 8 channels -> its LDS score row
#pragma unroll 1
        for (int s = 0; s < NSTG / 2; ++s) {
            TR_STAMP(gi, s, 0);                              // consumer: stage top
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TR_STAMP(gi, s, 1);                              // own LDS reads of the previous stage done
            __builtin_amdgcn_s_barrier();
            TR_STAMP(gi, s, 2);                              // barrier passed: the stage is in LDS
            TR_STAMP(gi, s, 3);
            const h16 *st = ring + cs_slot * STAGE_H + tid * 8;
            float d[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int l = s * R + r;
                h16x8 kk = l2d_ld8(st + r * NT * 8);
                if (l == u) l2d_st8(ku + coff, kk);
                kk = kk + l2d_ld8(kpt + l * 320);
                d[r] = ring_dot8(q8, kk);
            }
            if constexpr (R == 2) *reinterpret_cast<f32x2 *>(row + s * 2) = (f32x2){d[0], d[1]};
            else {
#pragma unroll
                for (int r = 0; r < R; ++r) row[s * R + r] = d[r];
            }
            TR_STAMP(gi, s, 4);                              // stage arithmetic issued
            cs_slot = (cs_slot + 1 == NS) ? 0 : cs_slot + 1;
        }
        // ---- per-head reduction over the HG threads of a head + the 1 x L softmax, on registers
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float sc[L];
#pragma unroll
        for (int l = 0; l < L; ++l) sc[l] = 0.f;
#pragma unroll 2
        for (int j = 0; j < HG; ++j) {
            const float *rr = sp + (gs + j) * LP;
#pragma unroll
            for (int l = 0; l < L; l += 4) {
                const f32x4 x = *reinterpret_cast<const f32x4 *>(rr + l);
                sc[l] += x[0]; sc[l + 1] += x[1]; sc[l + 2] += x[2]; sc[l + 3] += x[3];
            }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int l = 0; l < L; l += 4) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(blds + l);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[l + e] = sc[l + e] * scale + b4[e]; mx = fmaxf(mx, sc[l + e]); }
        }
        float den = 0.f;
#pragma unroll
        for (int l = 0; l < L; ++l) { sc[l] = __expf(sc[l] - mx); den += sc[l]; }
        const float inv = 1.0f / den;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every thread has read the partial rows: they may be overwritten
#pragma unroll
        for (int l = 0; l < L; l += 4) *reinterpret_cast<f32x4 *>(row + l) = (f32x4){sc[l] * inv, sc[l + 1] * inv, sc[l + 2] * inv, sc[l + 3] * inv};
        if (gi + 1 < ng) qn = l2d_ld8(qkv_b + (long long)(gi + 1) * (PB * 3 * C) + qoff);
        // ---- V stages: probabilities from the thread's own score row
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll 1
        for (int s = 0; s < NSTG / 2; ++s) {
            TR_STAMP(gi, s + NSTG / 2, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TR_STAMP(gi, s + NSTG / 2, 1);
            __builtin_amdgcn_s_barrier();
            TR_STAMP(gi, s + NSTG / 2, 2);
            TR_STAMP(gi, s + NSTG / 2, 3);
            const h16 *st = ring + cs_slot * STAGE_H + tid * 8;
            float pr[R];
            if constexpr (R == 2) { const f32x2 t = *reinterpret_cast<const f32x2 *>(row + s * 2); pr[0] = t[0]; pr[1] = t[1]; }
            else {
#pragma unroll
                for (int r = 0; r < R; ++r) pr[r] = row[s * R + r];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int l = s * R + r;
                h16x8 vv = l2d_ld8(st + r * NT * 8);
                if (l == u) l2d_st8(ku + slab + coff, vv);
                vv = vv + l2d_ld8(vpt + l * 320);
                ring_axpy8(o, pr[r], vv);
            }
            TR_STAMP(gi, s + NSTG / 2, 4);
            cs_slot = (cs_slot + 1 == NS) ? 0 : cs_slot + 1;
        }
        h16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (h16)o[e];
        l2d_st8(out_b + (long long)gi * (PB * C) + ooff, ov);
    }
}

template <int L, int HG, int R, int NS>
static void launch_ringlw(const TAttnArgs &a, const h16 *zero, int slots, hipStream_t s) {
    constexpr int NT = 320;
    constexpr size_t LDS = (size_t)NS * R * 8 * 40 * 16 + (size_t)(NT * (L + 4) + L) * sizeof(float) + (size_t)2 * L * 320 * sizeof(h16);
    static_assert(LDS <= 160 * 1024, "tattn ring geometry exceeds the CU's LDS");
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)tattn_stream_ringlw_kernel<HG, L, R, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) == hipSuccess)
            attr_done = true;
        else
            (void)hipGetLastError();
    }
    const int CH = a.C / 320;
    const int groups_per_unit = a.T / 8;
    const int total = a.N * CH * groups_per_unit;
    const int gpb = (total + slots - 1) / slots;
    const int bpu = (groups_per_unit + gpb - 1) / gpb;
    hipLaunchKernelGGL((tattn_stream_ringlw_kernel<HG, L, R, NS>), dim3(a.N * CH * bpu), dim3(384), LDS, s, a, zero, gpb, groups_per_unit);
}

#ifdef L2D_PROBES
template <int L, int HG, int R, int NS, int PB>
static void launch_ring_g(const TAttnArgs &a, const h16 *zero, int slots, hipStream_t s) {
    constexpr int NT = 40 * PB;
    constexpr size_t LDS = (size_t)NS * R * PB * 40 * 16 + (size_t)(NT * (L + 4) + L) * sizeof(float) + (size_t)2 * L * 320 * sizeof(h16);
    static_assert(LDS <= 160 * 1024, "tattn ring geometry exceeds the CU's LDS");
    static bool attr_done_dev[L2D_MAX_DEV] = {false};
    bool &attr_done = attr_done_dev[l2d_dev_ordinal()];
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)tattn_stream_ring_kernel<HG, L, R, NS, PB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) == hipSuccess)
            attr_done = true;
        else
            (void)hipGetLastError();
    }
    const int CH = a.C / 320;
    const int groups_per_unit = a.T / PB;
    const int total = a.N * CH * groups_per_unit;
    const int gpb = (total + slots - 1) / slots;
    const int bpu = (groups_per_unit + gpb - 1) / gpb;
    hipLaunchKernelGGL((tattn_stream_ring_kernel<HG, L, R, NS, PB>), dim3(a.N * CH * bpu), dim3(NT), LDS, s, a, zero, gpb, groups_per_unit);
}
#endif

template <int L, int HG>
static void launch_ring(const TAttnArgs &a, const h16 *zero, int cus, hipStream_t s) {
#ifdef L2D_PROBES
    // analysis builds: L2D_TATTN_RING=4 = the round-2 form (every wave issues its own DMAs; L <= 24), for A/B and stage stamps
    static int geo = -2;
    if (geo == -2) {
        const char *e = getenv("L2D_TATTN_RING");
        geo = e ? atoi(e) : -1;
    }
    if constexpr (L <= 16) { if (geo == 4) { launch_ring_g<L, HG, 4, 5, 8>(a, zero, cus, s); return; } }
    else if constexpr (L == 24) { if (geo == 4) { launch_ring_g<L, HG, 4, 4, 8>(a, zero, cus, s); return; } }
#endif
    if constexpr (L <= 16) {
        launch_ringlw<L, HG, 4, 5>(a, zero, cus, s);
    } else if constexpr (L == 24) {
        // 24 slots: PE rows 2 x 15 KB + score rows 35 KB leave room for 4 stages of 4 rows ((2 rows, 8 stages): 3 % slower)
        launch_ringlw<L, HG, 4, 4>(a, zero, cus, s);
    } else {
        launch_ringlw<L, HG, 2, 5>(a, zero, cus, s);
    }
}

bool l2d_tattn_ring_ok(const TAttnArgs &a, const void *zero) {
    return zero && (a.C == 320 || a.C == 640 || a.C == 1280) && (a.L == 16 || a.L == 12 || a.L == 24 || a.L == 40) && (a.T % 8 == 0) && a.H == 8;
}

int l2d_launch_tattn_ring(const TAttnArgs &a, const void *zero_page, hipStream_t s) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return L2D_ELAUNCH;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const h16 *zero = (const h16 *)zero_page;
    if (a.L == 16) {
        if (a.C == 320) launch_ring<16, 5>(a, zero, cus, s);
        else if (a.C == 640) launch_ring<16, 10>(a, zero, cus, s);
        else launch_ring<16, 20>(a, zero, cus, s);
    } else if (a.L == 12) {
        if (a.C == 320) launch_ring<12, 5>(a, zero, cus, s);
        else if (a.C == 640) launch_ring<12, 10>(a, zero, cus, s);
        else launch_ring<12, 20>(a, zero, cus, s);
    } else if (a.L == 24) {
        if (a.C == 320) launch_ring<24, 5>(a, zero, cus, s);
        else if (a.C == 640) launch_ring<24, 10>(a, zero, cus, s);
        else launch_ring<24, 20>(a, zero, cus, s);
    } else {
        if (a.C == 320) launch_ring<40, 5>(a, zero, cus, s);
        else if (a.C == 640) launch_ring<40, 10>(a, zero, cus, s);
        else launch_ring<40, 20>(a, zero, cus, s);
    }
    return L2D_OK;
}
