/*
 * l2d.h -- C ABI of libl2d_hip.so: the MI355X (gfx950) backend for Live2Diff's streaming UNet step.
 *
 * Drop-in boundary being replaced (reference, Python):
 *   - `stream.unet(sample, timestep, encoder_hidden_states=..., temporal_attention_mask=...,
 *      depth_sample=..., kv_cache=..., pe_idx=..., update_idx=...)`
 *        live2diff/pipeline_stream_animation_depth.py:456-466 (per frame), :355-365 (engine warm-up)
 *   - the accelerator object swapped in at live2diff/utils/wrapper.py:613-626, whose contract is
 *        live2diff/acceleration/tensorrt/engine.py:142-185 (UNet2DConditionModelDepthEngine.__call__)
 *        and whose runtime is live2diff/acceleration/tensorrt/utilities.py:247-294 (Engine.infer).
 * The reference has no FFI of its own (it is 100 % Python driving PyTorch / TensorRT); this header is
 * what a ctypes binding on the reference side binds instead of `Engine.infer`.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers (HBM) unless stated; sizes are ints; no torch types.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream). Nothing here synchronises.
 *   - every function returns 0 on success, a negative L2D_E* code otherwise; l2d_last_error() gives text.
 *   - activations are channels-last fp16: a tensor "[B,T,C]" is B*T rows of C contiguous halfs.
 *   - KV caches use the reference interchange layout [N,2,T,L,C] fp16
 *        (live2diff/animatediff/models/stream_motion_module.py:57-77).
 *
 * A UNet step is a static *plan*: an array of l2d_op records (one per kernel launch) that the host
 * builds once per (H,W,N,L) and that l2d_run_ops()/l2d_graph_* replay every frame.
 */
#ifndef L2D_H
#define L2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2D_ABI_VERSION 6

enum {
    L2D_OK = 0,
    L2D_EINVAL = -1,   /* bad argument / unsupported shape */
    L2D_ELAUNCH = -2,  /* HIP launch or runtime error */
    L2D_ENODEV = -3,   /* no gfx950 device */
};

/* op kinds ---------------------------------------------------------------------------------------
 * Field meaning per kind (p = pointers, i = ints, f = floats); unused fields must be 0.
 *
 * L2D_OP_IGEMM   implicit GEMM on MFMA: linear / 1x1 conv / 3x3 conv (stride 1|2, optional nearest-2x
 *                upsample folded into the gather, optional channel-concat of two inputs), fused epilogue.
 *                out[m][n] = epi( sum_k X[m][k] * W[n][k] )        (reference: nn.Linear / InflatedConv3d,
 *                resnet.py:57-65,112,141; attention.py:62,89; motion_module.py:182,207)
 *   p0 x1 [B,Hin,Win,C1] half   p1 x2 [.., C2] half or 0        p2 w packed [Nout][taps*CinP] half
 *   p3 bias[Nout] float or 0    p4 rowbias [*, ldrb] float or 0 p5 residual [M][ldr] half or 0
 *   p6 out [M][ldo] half
 *   i0 taps(1|9) i1 C1 i2 C2 i3 ldx1 i4 ldx2 i5 CinP i6 B i7 Hin i8 Win i9 Hout i10 Wout i11 stride
 *   i12 ups(0|1) i13 M i14 Nout i15 ldo i16 ldr i17 ldrb i18 rows_per_bias i19 epi (0 none, 1 GEGLU, 2 SiLU)
 *   i20 batch (grid.z) ; l0..l3 = per-batch element strides of x1, w, out, residual
 *   p7 16-byte zero page (padding source)   p8 split-K workspace [batch][S][M][round_up(Nout,4)] float
 *   i21 splitk (S, 1 = off: S > 1 adds the igemm_splitk_epilogue launch unless p11 is set)
 *   p11 (split-K only) int32 arrival counters, one per (batch, tile), ZERO before the launch and left zero by it: the
 *   reduction is fused -- the block that arrives last at a tile's counter sums the S partial tiles in the fixed order
 *   0..S-1 and runs the epilogue; p8 is then [batch][tile][S][tile_n * tile_m] float (whole tiles) and i22's tile must be
 *   explicit (1 or 2).  Results are bit-identical between runs (the order of the sum does not depend on the arrival order).
 *   i22 tile (0 auto, 1 = 128x128,
 *   2 = 64x64; + 16 = weight-tile-major block order: each XCD's L2 holds a band of output channels, for
 *   GroupNorm statistics of the OUTPUT, accumulated by the producer (replaces the consumer's L2D_OP_GN_STATS launch):
 *   p9 / p10 accumulators int64 [samples][G][2] of up to two consumer GroupNorms, or 0 ; i24 T (tokens per sample)
 *   i25 G ; i26 / i27 channels per group and channel offset of this tensor inside consumer 1's (concatenated) channel axis
 *   i28 / i29 the same for consumer 2.  Sums are deterministic fixed-point atomics: sum x in units of 2^-20, sum x^2 in
 *   units of 2^-12 (cannot overflow int64 for fp16 data at these sizes); the accumulators must be zero before the launch.
 *   Needs the LDS-staged epilogue (Nout % 8 == 0 ...) or split-K, and tiles that do not straddle samples (T % 128 == 0,
 *   or T % 64 == 0 with the 64x64 tile / split-K).
 *   i22 also: + 32 = direct register -> global epilogue (8-byte pieces) instead of the default one that
 *   transposes the tile through LDS and stores whole rows with 16 bytes per lane) i23 pipeline variant (igemm.hip launch_p)
 *
 * L2D_OP_GN_STATS / L2D_OP_GN_APPLY   GroupNorm over channels-last [B,T,C1(+C2)] (two-input = concat),
 *                optional SiLU (reference: InflatedGroupNorm resnet.py:68-76, F.silu :233,249)
 *   p0 x1 p1 x2|0 p2 partial[B][nchunk][G][2] float  (apply: p3 gamma half, p4 beta half, p5 out half)
 *   i0 B i1 T i2 C1 i3 C2 i4 ld1 i5 ld2 i6 G i7 nchunk i8 silu  f0 eps
 *   GN_APPLY with nchunk = 0: p6 = int64 [B][G][2] fixed-point accumulators filled by the producing igemm launches (above)
 *   GN_APPLY with nchunk = 0 and p6 = 0 (p2 unused): the one-launch form for small tensors -- a block holds all T rows of a band of whole
 *   groups (lcm(C / G, 8) <= 128 channels, <= 4 groups) in registers, statistics and apply in the same launch; refused (L2D_EINVAL) when
 *   T * band / 8 exceeds 16 vectors per thread of 256
 *
 * L2D_OP_LAYERNORM  p0 x [rows][ld] p1 gamma p2 beta p3 out [rows][C] ; i0 rows i1 C i2 ldx i3 ldo; f0 eps
 *
 * L2D_OP_FLASH_ATTN  softmax(QK^T/sqrt(d))V on MFMA (spatial self / text cross attention;
 *                reference call sites attention.py:243,250-255)
 *   p0 q [B][Tq][ldq] p1 k [B][Tk][ldk] p2 vt [B][H*d][ldvt] (V transposed) p3 out [B][Tq][ldo]
 *   i0 B i1 H i2 d i3 Tq i4 Tk i5 ldq i6 ldk i7 ldvt i8 ldo ; l0 q batch stride l1 k l2 vt l3 out
 *   p4 16-byte zero page (DMA source of keys beyond Tk; 0 selects the register-staged kernel)
 *   i9 variant: 0 auto (LDS-DMA ring kernel, flash_attn_ring.hip), 1 register-staged kernel (flash_attn.hip: the fallback for
 *   K / V^T operands that are not 16-byte aligned), 2 / 3 ring kernel with 32 / 16 query rows per wave, 4 ring kernel with 32
 *   rows per wave and the software-pipelined key-tile loop (d <= 48; bit-identical to 2; what auto picks where 2 was picked
 *   before; larger d run 2).  V^T columns in [Tk, ldvt) may hold anything.
 *
 * L2D_OP_TATTN_STREAM  fused streaming temporal attention with multi-timestep KV-cache
 *                (reference stream_motion_module.py:99-213)
 *   p0 qkv [N*T][3C] half p1 cache [N,2,T,L,C] half (in-place) p2 q_pe p3 k_pe p4 v_pe [maxlen][C] half
 *   p5 pe_idx [N][L] int64 p6 update_idx [N] int64 p7 bias [N][L] half p8 out [N*T][C] half
 *   p9 16-byte zero page (DMA source of masked slots; required by the ring kernel, may be 0 otherwise)
 *   i0 N i1 T i2 C i3 L i4 H i5 variant (0 auto: the loader-wave LDS-DMA ring kernel (tattn_ring.hip) for C in {320,640,1280},
 *   L in {12,16,24,40}, T % 8 == 0, else register-resident (L <= 16) / chunked; 1 register-resident, 2/3 chunked CH=8/4,
 *   13 = the ring kernel, refused when the shape does not fit it)
 *
 * L2D_OP_TATTN_WARMUP  bidirectional warm-up temporal attention + cache fill
 *                (reference motion_module.py:469-530)
 *   p0 qkv [F*T][3C] p1 cache_row [2,T,L,C] p2 q_pe p3 k_pe p4 v_pe p8 out [F*T][C]
 *   i0 F i1 T i2 C i3 L i4 H
 *
 * L2D_OP_SKINNY_LINEAR  out[m][n] = act( sum_k A[m][k] W[n][k] + b[n] ), m < 8 (time embedding path,
 *                reference unet_depth_streaming.py:499-505, resnet.py:238)
 *   p0 A half [M][K] p1 W half [Nout][K] p2 bias float p3 out (half or float) ; i0 M i1 K i2 Nout
 *   i3 silu_out i4 out_is_float i5 ldo
 * L2D_OP_TIMESTEP_EMBED  p0 timesteps (int64 [N]) p1 out half [N][dim]; i0 N i1 dim
 * L2D_OP_NCHW_TO_NHWC  p0 in half [B][C][HW] p1 out half [B][HW][Cpad]; i0 B i1 C i2 HW i3 Cpad
 * L2D_OP_NHWC_TO_NCHW  p0 in half [B][HW][ld] p1 out half [B][C][HW]; i0 B i1 C i2 HW i3 ld
 *   both: i4 element map, fp16 rounding after each step: 0 copy, 1 (x + f1) * f0, 2 tanh(x / 3) * 3, 3 x * f0 + f1
 *   (the input / output scalings of diffusers' EncoderTiny / DecoderTiny, reference swap point wrapper.py:468-470)
 * L2D_OP_LCM_STEP   x0 = c_out*(x - beta*eps)/alpha + c_skip*x  (reference pipeline :387-401)
 *   p0 x p1 eps p2 scal float [N][4]={alpha,beta,c_skip,c_out} p3 x0 ; i0 N i1 per_sample_elems
 * L2D_OP_COPY       p0 src p1 dst ; l0 bytes   (device-to-device, on the stream)
 *
 * Per-frame pipeline glue on the device (stream_glue.hip):
 * L2D_OP_RING_UPDATE  update_attn_bias (reference pipeline_stream_animation_depth.py:416-438), in place:
 *   p0 bias [N][L] half (0 / -inf) p1 pe_idx [N][L] int64 p2 update_idx [N] int64 p3 frame counter uint64* or 0
 *   (incremented by one); i0 N i1 L i2 sink (= warm-up frames)
 * L2D_OP_STREAM_SHIFT scheduler_step_batch + stream-batch shift register (reference :387-401, :590-601):
 *   p0 x_t [N][per] half (in: the batch the UNet just saw; out: rows 1..N-1 = next frame's buffer)
 *   p1 eps [N][per] half (UNet output) p2 scal [N][4] float {alpha, beta, c_skip, c_out} p3 noise [N-1][per] half or 0
 *   p4 x0_out [per] half (x0 prediction of the last row) p5 depth [N][per] half or 0 (row i+1 <- row i)
 *   i0 N (<= 8) i1 per (elements per row)
 * L2D_OP_RANDN      p0 out [n] half, standard normal (Philox4x32-10 + Box-Muller) p1 frame counter uint64* or 0
 *   l0 n l1 seed l2 offset (in Philox blocks of 4); element i = normal i%4 of block offset + frame*ceil(n/4) + i/4
 *
 * Depth-path glue (glue.hip; reference pipeline_stream_animation_depth.py:553,560-567):
 * L2D_OP_RESIZE_BILINEAR  F.interpolate(mode="bilinear", align_corners=False) on fp16 planes:
 *   p0 in [planes][Hin][Win] half p1 out [planes][Hout][Wout] half ; i0 planes i1 Hin i2 Win i3 Hout i4 Wout
 * L2D_OP_MINMAX  min and max of a fp16 tensor, left on the device: p0 x half (16-byte aligned) p1 scratch float [2*nb]
 *   p2 out float[2] = {min, max} ; l0 n ; i0 nb (blocks of the partial pass, <= 1024)
 * L2D_OP_DEPTH_NORM_RESIZE  ((d - min) / (max - min)) -> 3 channels -> * 2 - 1 -> bilinear resize, fp16 rounding after each
 *   reference tensor op:  p0 depth [B][Hd][Wd] half p1 {min, max} float[2] p2 out [B][3][H][W] half ; i0 B i1 Hd i2 Wd i3 H i4 W
 *
 * Depth detector (DPT-Hybrid; reference depth_utils.py:11-32): its GEMM-shaped work runs on L2D_OP_IGEMM
 * (i30 = 1: TF-"SAME" low-side padding 0 for stride-2 3x3 convs; epi 5 = GELU) / L2D_OP_FLASH_ATTN (d = 64) /
 * L2D_OP_GN_APPLY (i8 activation: 0 none, 1 SiLU, 2 ReLU, 3 ReLU(norm(x) + p7 residual)) / L2D_OP_LAYERNORM /
 * L2D_OP_SKINNY_LINEAR (l0 = row stride of A, 0 = dense); the rest:
 * L2D_OP_STEM7X7   weight-standardised 7x7 stride-2 SAME conv, 3 -> 64: p0 image [B,3,H,W] half (NCHW) p1 w [64][3][7][7] half
 *   p2 out [B,ceil(H/2),ceil(W/2),64] half ; i0 B i1 H i2 W
 * L2D_OP_RESAMPLE_NHWC  channels-last p0 in [B,H,W,C] -> p1 out ; i0 B i1 H i2 W i3 C (% 8) i4 mode: 0 = 3x3 stride-2 max pool
 *   (SAME), 1 = stride-2 subsample, 2 = bilinear x2 upsample with align_corners=True
 * L2D_OP_EW        s = p0 (+ p1); p2 = s (if given); p3 = relu(s) (if given) ; l0 n halfs (% 8)
 *
 * L2D_OP_ROWGEMM   token-row GEMM for the transformer linear layers, with the normalisation in front of them fused:
 *                out[m][n] = epi( sum_k T(x)[m][k] * W[n][k] ),  K % 64 == 0, K <= 2048  (rowgemm.hip; reference: nn.LayerNorm /
 *                GroupNorm + nn.Linear, attention.py:57-62,89,102-110,173-205; motion_module.py:181-182,207,273-279,355-361)
 *   p0 x [M][ldx] half   p1 w half, packed in MFMA-fragment order [Nout/32][K/16][64 lanes][8] (ops.pack_rowgemm; GEGLU: rows
 *   permuted so that a 32-row tile holds 8 value / 8 gate / 8 value / 8 gate rows of 16 output channels)
 *   p2 bias float [Nout] (packed order) or 0   p3 residual [M][ldr] half or 0   p4 out [M][ldo] half
 *   (the affine part of the prologue norm is folded into p1 / p2 at pack time: the kernel takes no gamma / beta; p5, p6 unused)
 *   p7 prologue 2: int64 [samples][G][2] fixed-point statistics of x
 *   p8 outT half: output of the LAST i15 weight tiles, stored transposed [sample][channel][ldt] (V^T for the flash kernel)
 *   p9 / p10, i24..i29: GroupNorm statistics of the output for up to two consumers, exactly as L2D_OP_IGEMM
 *   i0 M i1 K i2 Nout (packed rows, % 32) i3 ldx i4 ldo i5 ldr i6 epi (0 none, 1 GEGLU: Nout / 2 output columns)
 *   i7 prologue (0 none, 1 LayerNorm over K, 2 GroupNorm normalisation from p7)   (i8 unused)
 *   i9 T tokens per sample (prologue 2, transposed output, output statistics: T % (32 MT) == 0) i10 G groups of prologue 2
 *   i12 NW waves per block (1..8; 1..5 when NT >= 3) i13 NT weight tiles per wave (1..4) i14 MT token tiles per block
 *   (1 | 2 | 4; MT >= 2 needs NT <= 2, MT = 4 needs K = 320): a block computes 32 MT tokens x 32 NW NT packed rows;
 *   (Nout / 32) % (NW NT) == 0 (ops.rowgemm_schedule)
 *   i15 trailing weight tiles stored transposed to p8 (% (NW NT)) i16 ldt i17 block order (1 = weight-band major per XCD)
 *   l0 elements between samples in p8 ; f0 eps of the prologue norm
 *
 * L2D_OP_PCONV     3x3 stride-1 pad-1 convolution with the haloed activation patch resident in LDS (pconv.hip; reference
 *                InflatedConv3d resnet.py:57-65 as used by ResnetBlock3D :194,214): same result as L2D_OP_IGEMM with taps = 9,
 *                stride 1, no upsample, epi 0; the activations are fetched once per 64-channel chunk instead of once per tap
 *   p0 x1 [B,H,W,C1] half   p1 x2 [B,H,W,C2] half or 0 (channel concat)   p2 w packed [Nout][9 * CinP] half (ops.pack_conv3x3)
 *   p3 bias float [Nout] or 0   p4 rowbias float [*][ldrb] or 0 (row = first token of the sample / i18)   p5 residual [M][ldr] half
 *   or 0   p6 out [M][ldo] half   p7 16-byte zero page   p9 / p10, i24..i29 GroupNorm statistics of the output as L2D_OP_IGEMM
 *   i1 C1 i2 C2 (both % 64) i3 ldx1 i4 ldx2 i5 CinP (= C1 + C2) i6 B i7 H i8 W i9 / i10 patch height / width (8x16, 8x8 or 4x8;
 *   H % PH == 0, W % PW == 0) i11 block order (1 = weight-tile major) i14 Nout (% 64) i15 ldo i16 ldr i17 ldrb i18 rows_per_bias
 *
 * L2D_OP_WSGEMM    weight-streaming GEMM for the levels with few tokens (M = N * T <~ 1k): linear layers and 3x3 stride-1 pad-1
 *                convs with 128-token row tiles; every weight byte is fetched by one wave per row tile, straight into registers
 *                (wsgemm.hip; reference: resnet.py:194,214 via InflatedConv3d :57-65; attention.py:173-205,258;
 *                motion_module.py:360; stream_motion_module.py:99-147).  out[m][n] = epi( sum_k LN?(x)[m][k] * W[n][k] )
 *   p0 x1 [M][ldx1] half (conv: [B,H,W,C1])   p1 x2 [M][ldx2] half or 0 (channel concat)   p2 w half, MFMA-fragment order
 *   [Nout/32][taps*CinP/16][64 lanes][8] with k = tap * CinP + channel (ops.pack_wsgemm / pack_wsgemm_conv3x3; GEGLU rows permuted
 *   as for L2D_OP_ROWGEMM)   p3 bias float [Nout] (packed order) or 0   p4 rowbias float [*][ldrb] or 0 (row = token / i18)
 *   p5 residual [M][ldr] half or 0   p6 out [M][ldo] half   p7 zero region, >= 2 * CinP + 256 zero bytes (padding / ragged rows)
 *   p8 outT half: output of the LAST i21 weight tiles, stored transposed [sample][channel][ldt] (V^T; T % 128 == 0)
 *   p9 / p10, i24..i29 GroupNorm statistics of the output as L2D_OP_IGEMM (a tile may span samples: T % 32 == 0)
 *   p11 split-K arrival counters (int32, one per (channel tile, row tile), zero before and after)   p12 split-K workspace float
 *   [tiles][S][128 * 32 NW NT + 256]   p13 colsum float [Nout]: sum_k fp16(W'[n][k]) for the LayerNorm fold (i20 = 1)
 *   i0 taps (1 | 9) i1 C1 i2 C2 (% 64) i3 ldx1 i4 ldx2 i5 CinP (= C1 + C2) i6 B i7 H i8 W (conv; W >= 8, M = B H W)
 *   i9 NW consumer waves (1..10; 9 / 10 since round 6: 3 / 3 / 2 / 2 consumers per SIMD, one block per CU) i10 NT weight tiles per wave (1 | 2; 2: NW <= 4) i11 NL loader waves (1 | 2) i12 S K slices
 *   (fused reduction: the last arriving block sums the S slabs in order 0..S-1 -- bit-repeatable -- and runs the epilogue)
 *   i13 M i14 Nout (packed rows, % 32; (Nout / 32) % (NW NT) == 0) i15 ldo i16 ldr i17 ldrb i18 rows_per_bias (% 32)
 *   i19 epi (0 none, 1 GEGLU: Nout / 2 output columns) i20 pro (0 none, 1 LayerNorm over K folded: gamma / beta live in p2 / p3,
 *   out = rstd (acc - mean colsum) + bias with the row statistics taken in the kernel) i21 trailing weight tiles stored transposed
 *   i22 ldt i23 non-temporal weight loads (single row tile) i30 T tokens per sample (p8)   l0 elements between samples in p8
 *   f0 eps of the LayerNorm
 *
 * L2D_OP_CCONV     3x3 stride-1 pad-1 convolution with the haloed activation patch resident in LDS AND the weights streamed straight into
 *                registers (cconv.hip, round 6; reference InflatedConv3d resnet.py:57-65 as used by ResnetBlock3D :194,214 and by
 *                Upsample3D :94-127): same result as L2D_OP_IGEMM with taps = 9, stride 1 (optionally the nearest-x2 up-sampling folded into
 *                the gather), epi 0.  A block owns an 8 x 16 patch of OUTPUT pixels x 64 CG channels x one K slice of whole 64-channel chunks.
 *   p0 x1 [B,Hs,Ws,C1] half   p1 x2 [B,Hs,Ws,C2] half or 0 (channel concat)   p2 w half: the weight streams of ops.pack_cconv(weight, KG)
 *   ([Nout/64][KG][chunk][tap][u < 4/KG][half 2][64 lanes][8], k step kk = u KG + kg; + 9 k steps (18 KB) of zero padding)
 *   p3 bias float [Nout] or 0   p4 rowbias float [*][ldrb] or 0 (row = first token of the sample / i18)   p5 residual [M][ldr] half or 0
 *   p6 out [M][ldo] half   p7 16-byte zero page   p9 / p10, i24..i29 GroupNorm statistics of the OUTPUT as L2D_OP_IGEMM
 *   p11 split-K counters (int32, TWO per (channel tile, patch): tickets and done; zero before and after)   p12 split-K workspace float
 *   [tiles][S][128 * 64 CG]   (S > 1: the block with the last ticket sums the S partial tiles in the fixed order 0..S-1 with its own
 *   registers at its own position -- bit-repeatable; it neither publishes nor re-reads its own tile)
 *   p13 / p14 / p15, i20 = 1, i21 G, f0 eps: the conv of silu(GroupNorm(x1 | x2)) -- statistics int64 [B][G][2] as L2D_OP_GN_APPLY with
 *   nchunk = 0, gamma / beta half [C1 + C2]; the normalisation runs in the loader waves (no GroupNorm launch, no normalised tensor)
 *   i1 C1 i2 C2 (both % 64) i3 ldx1 i4 ldx2 i5 CinP (= C1 + C2) i6 B i7 H i8 W (OUTPUT resolution: H % 8 == 0, W % 16 == 0; the input is
 *   [B, H >> i13, W >> i13, C]) i9 CG i10 KG ((2, 2) or (1, 4): CG KG = 4 compute waves) i11 NLD loader waves (1 | 2 | 4) i12 S K slices
 *   (<= CinP / 64) i13 ups (0 | 1: nearest x2 of Upsample3D) i14 Nout (% 64 CG) i15 ldo i16 ldr i17 ldrb i18 rows_per_bias
 *
 * L2D_OP_ROWCHAIN  token-resident tail of a transformer block, ONE launch (rowchain.hip; reference attention.py:243-270,125-133;
 *                motion_module.py:401-435,290-297):   h2 = to_out(a) + res1;  h3 = FF2(GEGLU(LayerNorm(h2))) + h2;
 *                out = proj_out(h3) + res2.   C = 320 (the level whose M / 32 blocks fill the chip), M % 32 == 0.
 *   p0 a [M][lda] half (attention output)   p1 res1 [M][ldr1] half (residual stream)   p2 res2 [M][ldr2] half (block input)
 *   p3 out [M][ldo] half   p4 / p5 to_out weights (L2D_OP_ROWGEMM fragment order, ops.pack_rowgemm) / fp32 bias
 *   p6 / p7 GEGLU projection (LayerNorm gamma / beta folded, value / gate rows interleaved as for L2D_OP_ROWGEMM epi 1) / bias
 *   p8 / p11 FF2 weights / bias   p12 / p13 proj_out weights / bias   p9 / p10, i24..i29 GroupNorm statistics of `out` as
 *   L2D_OP_IGEMM (T % 32 == 0)
 *   i0 M i1 C i2 lda i3 ldr1 i4 ldr2 i5 ldo i6 = 0   f0 eps of the LayerNorm
 *   i6 = 1: HEAD SEGMENT, two dependent layers in one launch (attention.py:102-110,221-250; motion_module.py:273-279,401-427):
 *                h = A(norm?(x)) + bA (+ resA) -> p2;   out = B(LayerNorm(h)) + bB, i7 passes of C packed rows each
 *   p0 x [M][i2] half   p1 resA [M][i3] half or 0   p2 h out [M][i4] half   p3 out [M][i5] half: the first (i7 - i8) passes, C columns each
 *   p4 / p5 layer A weights (fragment order; a GroupNorm's affine folded) / fp32 bias   p6 / p7 layer B weights (LayerNorm folded) /
 *   fp32 bias or 0   p8 outT half: with i8 = 1 the LAST pass is stored transposed [sample][channel][i9] (V^T), l0 elements between
 *   samples   p14 GroupNorm prologue on x: int64 [samples][G][2] fixed-point statistics filled by x's producers (as L2D_OP_ROWGEMM
 *   prologue 2), or 0   i7 passes (1 | 3) i8 transposed last pass (needs i7 = 3) i9 ldt i10 T tokens per sample i11 G
 *   f0 eps of the LayerNorm f1 eps of the GroupNorm
 */
enum {
    L2D_OP_IGEMM = 1,
    L2D_OP_GN_STATS = 2,
    L2D_OP_GN_APPLY = 3,
    L2D_OP_LAYERNORM = 4,
    L2D_OP_FLASH_ATTN = 5,
    L2D_OP_TATTN_STREAM = 6,
    L2D_OP_TATTN_WARMUP = 7,
    L2D_OP_SKINNY_LINEAR = 8,
    L2D_OP_TIMESTEP_EMBED = 9,
    L2D_OP_NCHW_TO_NHWC = 10,
    L2D_OP_NHWC_TO_NCHW = 11,
    L2D_OP_LCM_STEP = 12,
    L2D_OP_COPY = 13,
    L2D_OP_RING_UPDATE = 14,
    L2D_OP_STREAM_SHIFT = 15,
    L2D_OP_RANDN = 16,
    L2D_OP_RESIZE_BILINEAR = 17,
    L2D_OP_MINMAX = 18,
    L2D_OP_DEPTH_NORM_RESIZE = 19,
    L2D_OP_STEM7X7 = 20,
    L2D_OP_RESAMPLE_NHWC = 21,
    L2D_OP_EW = 22,
    L2D_OP_ROWGEMM = 23,
    L2D_OP_PCONV = 24,
    L2D_OP_WSGEMM = 25,
    L2D_OP_ROWCHAIN = 26,
    L2D_OP_CCONV = 27,
};

typedef struct l2d_op {
    int32_t kind;
    int32_t tag;          /* free for the host (plan index / layer id); echoed in error messages */
    void *p[16];
    int32_t i[32];
    int64_t l[4];
    float f[4];
} l2d_op;

/* library / device ------------------------------------------------------------------------------ */
int l2d_abi_version(void);
const char *l2d_last_error(void);
/* 0 if the current HIP device is a gfx950 part; fills name (<= 63 chars) when non-NULL. */
int l2d_device_check(char *name, int name_len);

/* Validate-only mode: when on, l2d_run_ops checks every op's arguments exactly as a real launch would and
 * returns without touching the device (used by the CPU test-suite to check whole plans). */
int l2d_set_dry_run(int on);

/* execution ------------------------------------------------------------------------------------- */
/* Launch ops[0..n) in order on `stream`. */
int l2d_run_ops(const l2d_op *ops, int n, void *stream);
/* Capture ops[0..n) into a hipGraph on `stream` (stream capture) and instantiate it. */
int l2d_graph_create(const l2d_op *ops, int n, void *stream, void **graph_out);
int l2d_graph_launch(void *graph, void *stream);
int l2d_graph_destroy(void *graph);

/* timing helper for bench.py: hipEvent pair around `reps` runs of the ops on `stream`; returns ms. */
int l2d_time_ops(const l2d_op *ops, int n, void *stream, int reps, float *ms_out);

/* per-launch timing inside the op sequence (an event in front of every op, `reps` passes): us_out[n] = mean microseconds
 * from op i's start marker to op i+1's (launch + the gap behind it).  For tools (schedule tuning, in-frame breakdowns). */
int l2d_time_each(const l2d_op *ops, int n, void *stream, int reps, float *us_out);

/* HBM copy microbenchmark (device-to-device float4 copy kernel), used to state the measured HBM peak
 * next to the datasheet number: copies `bytes` from src to dst `reps` times, returns GB/s (read+write). */
int l2d_copy_bench(const void *src, void *dst, int64_t bytes, int reps, void *stream, float *gbps_out);

/* HBM read probe: streaming 16-byte loads, `unroll` (1,2,4,8,16) independent loads in flight per thread,
 * 256 x blocks_per_cu blocks of 256 threads; returns GB/s read. */
int l2d_read_bench(const void *src, void *sink, int64_t bytes, int unroll, int blocks_per_cu, int reps, void *stream,
                   float *gbps_out);

#ifdef __cplusplus
}
#endif
#endif /* L2D_H */
