#!/usr/bin/env python
"""Benchmark of the hot path: the per-frame streaming UNet step of Live2Diff on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Unit of work ("step") = one streaming UNet forward over the denoising batch = one output frame
(reference call site pipeline_stream_animation_depth.py:456-466).  Workload = BASELINE.json configs[1]:
512x512 image (64x64 latent), 2 denoise steps (t = [399, 199]), window L = 8 sink + 8 rolling = 16,
SD-1.5 widths (1 277.7 M parameters, random-init key-hashed fp16 weights), synthetic N(0,1) inputs,
KV caches pre-filled N(0,1), steady-state ring buffer (all 16 slots unmasked: worst case for bandwidth).
Inputs and weights are resident in HBM before the timed region; every timed step includes the boundary's
input copies and the host ring-buffer update + upload, exactly as the pipeline drives it.

N GPUs = N independent streams (weak scaling), weights broadcast once over RCCL; `value` = aggregate frames/s.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0    # dense fp16/bf16 MFMA
MFMA_KERNELS = ("igemm_kernel", "rowgemm_kernel", "pconv_kernel", "cconv_kernel", "wsgemm_kernel", "rowchain_kernel", "flash_attn_kernel")   # kernel families priced against the MFMA roofline


# (height, width, denoise steps, window L) of the BASELINE.json configurations the GPU leg can run
WORKLOAD_NAMES = {(256, 256, 1, 12): "BASELINE configs[0]", (512, 512, 2, 16): "BASELINE configs[1]",
                  (512, 768, 2, 24): "BASELINE configs[2]", (512, 512, 4, 16): "BASELINE configs[3] (one of the 8 streams per GPU)",
                  (576, 1024, 2, 40): "BASELINE configs[4]"}


def op_work(op, kinds):
    """Algorithmic work of one plan op: (flops, bytes) from the record's own fields."""
    k = op.kind
    i = op.i
    if k == kinds.OP_IGEMM:
        taps, C1, C2, M, Nout, batch = i[0], i[1], i[2], i[13], i[14], max(1, i[20])
        nreal = Nout
        flops = 2.0 * M * nreal * taps * (C1 + C2) * batch
        byts = 2.0 * batch * (M * (C1 + C2) + nreal * taps * (C1 + C2) + M * (Nout // 2 if i[19] == 1 else Nout))
        return flops, byts
    if k == kinds.OP_PCONV:
        B, H, W_, C, Nout = i[6], i[7], i[8], i[1] + i[2], i[14]
        M = B * H * W_
        return 2.0 * M * Nout * 9 * C, 2.0 * (M * C + Nout * 9 * C + M * Nout)
    if k == kinds.OP_CCONV:
        B, H, W_, C, Nout, ups = i[6], i[7], i[8], i[1] + i[2], i[14], i[13]
        M = B * H * W_
        return 2.0 * M * Nout * 9 * C, 2.0 * ((M >> (2 * ups)) * C + Nout * 9 * C + M * Nout)
    if k == kinds.OP_ROWGEMM:
        M, K, Nout = i[0], i[1], i[2]
        return 2.0 * M * Nout * K, 2.0 * (M * K + Nout * K + M * (Nout // 2 if i[6] == 1 else Nout))
    if k == kinds.OP_WSGEMM:
        taps, C, M, Nout = i[0], i[1] + i[2], i[13], i[14]
        return 2.0 * M * Nout * taps * C, 2.0 * (M * C + Nout * taps * C + M * (Nout // 2 if i[19] == 1 else Nout))
    if k == kinds.OP_ROWCHAIN:
        M, C = i[0], i[1]
        if i[6] == 1:                     # head segment: A (C x C) + B (passes x C x C); in: x (+ resA); out: h, passes x C columns
            nb = i[7]
            return 2.0 * M * (1 + nb) * C * C, 2.0 * (M * C * (2 + nb + (1 if op.p[1] else 0)) + (1 + nb) * C * C)
        # tail: to_out C x C, GEGLU C x 8C, FF2 4C x C, proj_out C x C; in: a, res1, res2; out
        return 2.0 * M * 14 * C * C, 2.0 * (4 * M * C + 14 * C * C)
    if k == kinds.OP_FLASH_ATTN:
        B, H, d, Tq, Tk = i[0], i[1], i[2], i[3], i[4]
        return 4.0 * B * H * Tq * Tk * d, 2.0 * B * H * d * (2 * Tq + 2 * Tk)
    if k in (kinds.OP_TATTN_STREAM,):
        N, T, C, L = i[0], i[1], i[2], i[3]
        # K+V slabs read once, new row written, q in / out:  4*N*T*L*C + 8*N*T*C bytes (fp16)
        return 4.0 * N * T * L * C, 4.0 * N * T * L * C + 8.0 * N * T * C
    if k in (kinds.OP_GN_STATS,):
        B, T, C = i[0], i[1], i[2] + i[3]
        return 3.0 * B * T * C, 2.0 * B * T * C
    if k in (kinds.OP_GN_APPLY,):
        B, T, C = i[0], i[1], i[2] + i[3]
        return 4.0 * B * T * C, 4.0 * B * T * C
    if k == kinds.OP_LAYERNORM:
        return 8.0 * i[0] * i[1], 4.0 * i[0] * i[1]
    return 0.0, 0.0


KIND_NAMES = {1: "igemm_kernel", 2: "gn_stats_kernel", 3: "gn_apply_kernel", 4: "layernorm_kernel", 5: "flash_attn_kernel",
              6: "tattn_stream_kernel", 7: "tattn_warmup_kernel", 8: "skinny_linear_kernel", 9: "timestep_embed_kernel",
              10: "nchw_to_nhwc_kernel", 11: "nhwc_to_nchw_kernel", 12: "lcm_step_kernel", 13: "copy", 23: "rowgemm_kernel", 24: "pconv_kernel", 25: "wsgemm_kernel",
              26: "rowchain_kernel", 27: "cconv_kernel"}


def per_kernel_breakdown(unet, reps=5):
    """Per-kernel-family time of one frame, measured live with HIP events (l2d_time_ops records hipEvents on the
    stream the kernels are launched on) by replaying each family's launches of the plan back to back."""
    from live2diff_amd import _lib
    st = unet._plans["stream"]
    groups = {}
    for j in range(len(st.pl)):
        groups.setdefault(st.pl[j].kind, []).append(st.pl[j])
    rows = {}
    for kind, ops_ in groups.items():
        pl = _lib.OpList()
        for op in ops_:
            c = _lib.L2dOp()
            import ctypes
            ctypes.memmove(ctypes.byref(c), ctypes.byref(op), ctypes.sizeof(_lib.L2dOp))
            pl.append(c)
        pl.time_ms(1)
        ms = pl.time_ms(reps)
        fl = sum(op_work(o, _lib)[0] for o in ops_)
        by = sum(op_work(o, _lib)[1] for o in ops_)
        rows[KIND_NAMES.get(kind, str(kind))] = dict(launches=len(ops_), ms=ms, flops=fl, bytes=by,
                                                     avg_us=1e3 * ms / len(ops_))
    return rows


def tattn_variant_sweep(unet, reps=5, variants=(1, 13)):
    """A/B of the streaming temporal-attention kernel variants on this frame's 40 launches (same process,
    interleaved): ms per frame for variant 1 (register resident) and 13 (the LDS-DMA ring kernel, the default).
    (The analysis-only probe variants 8-12 exist in `make PROBES=1` builds only.)"""
    import ctypes

    from live2diff_amd import _lib
    st = unet._plans["stream"]
    src = [st.pl[j] for j in range(len(st.pl)) if st.pl[j].kind == _lib.OP_TATTN_STREAM]
    lists = {}
    for v in variants:
        pl = _lib.OpList()
        for op in src:
            c = _lib.L2dOp()
            ctypes.memmove(ctypes.byref(c), ctypes.byref(op), ctypes.sizeof(_lib.L2dOp))
            c.i[5] = v
            pl.append(c)
        lists[v] = pl
    out = {v: [] for v in lists}
    for _ in range(3):
        for v, pl in lists.items():
            if out[v] is None:
                continue
            try:
                pl.time_ms(1)
                out[v].append(pl.time_ms(reps))
            except _lib.L2DError:       # a forced variant some level of this resolution cannot take (ring: T % 8, SD widths)
                out[v] = None
    return {f"v{v}": (round(min(t), 4) if t else None) for v, t in out.items()}


def op_dims(op, kinds):
    i = op.i
    if op.kind == kinds.OP_IGEMM:
        return (f"taps{i[0]} M{i[13]} N{i[14]} K{i[0] * (i[1] + i[2])} Kp{i[0] * i[5]} s{i[11]} u{i[12]} e{i[19]} "
                f"b{max(1, i[20])} S{max(1, i[21])} t{i[22] & 15} v{i[23]} o{i[22] >> 4}")
    if op.kind == kinds.OP_PCONV:
        return f"B{i[6]} H{i[7]} W{i[8]} C{i[1] + i[2]} N{i[14]} patch{i[9]}x{i[10]} o{i[11]}"
    if op.kind == kinds.OP_CCONV:
        return f"B{i[6]} H{i[7]} W{i[8]} C{i[1] + i[2]} N{i[14]} u{i[13]} cg{i[9]} kg{i[10]} l{i[11]} S{max(1, i[12])}"
    if op.kind == kinds.OP_ROWGEMM:
        i = op.i
        return f"M{i[0]} N{i[2]} K{i[1]} e{i[6]} p{i[7]} w{i[12]} t{i[13]} m{i[14]} tr{i[15]}"
    if op.kind == kinds.OP_WSGEMM:
        return f"taps{i[0]} M{i[13]} N{i[14]} K{i[0] * (i[1] + i[2])} e{i[19]} p{i[20]} w{i[9]} t{i[10]} l{i[11]} S{max(1, i[12])} tr{i[21]} nt{i[23]}"
    if op.kind == kinds.OP_ROWCHAIN:
        return f"M{i[0]} C{i[1]} " + (f"head p{i[7]} tr{i[8]} gn{int(bool(op.p[14]))} res{int(bool(op.p[1]))}" if i[6] == 1 else "tail")
    if op.kind == kinds.OP_FLASH_ATTN:
        return f"B{i[0]} H{i[1]} d{i[2]} Tq{i[3]} Tk{i[4]}"
    if op.kind in (kinds.OP_TATTN_STREAM, kinds.OP_TATTN_WARMUP):
        return f"N{i[0]} T{i[1]} C{i[2]} L{i[3]}"
    if op.kind in (kinds.OP_GN_STATS, kinds.OP_GN_APPLY):
        return f"B{i[0]} T{i[1]} C{i[2] + i[3]} nchunk{i[7]}"
    if op.kind == kinds.OP_LAYERNORM:
        return f"rows{i[0]} C{i[1]}"
    return ""


def dump_plan(unet, path):
    """The stream plan as CSV (index, kernel, dims, flops, bytes, launches): tools/frame_trace.py joins it with a
    rocprofv3 kernel trace to get the IN-FRAME duration of every launch (cold weights, real neighbours)."""
    from live2diff_amd import _lib
    st = unet._plans["stream"]
    with open(path, "w") as f:
        f.write("idx,kernel,dims,flops,bytes,dispatches\n")
        for j in range(len(st.pl)):
            op = st.pl[j]
            fl, by = op_work(op, _lib)
            nd = 0 if op.kind == _lib.OP_COPY else (2 if op.kind == _lib.OP_IGEMM and op.i[21] > 1 and not op.p[11] else 1)
            f.write(f"{j},{KIND_NAMES.get(op.kind, op.kind)},{op_dims(op, _lib)},{fl:.0f},{by:.0f},{nd}\n")


def per_op_table(unet, path, reps=10):
    """Every launch of the frame timed on its own (HIP events, `reps` back-to-back runs): CSV for tuning."""
    import ctypes

    from live2diff_amd import _lib
    st = unet._plans["stream"]
    with open(path, "w") as f:
        f.write("tag,kernel,dims,us,tflops,gbps\n")
        for j in range(len(st.pl)):
            op = st.pl[j]
            c = _lib.L2dOp()
            ctypes.memmove(ctypes.byref(c), ctypes.byref(op), ctypes.sizeof(_lib.L2dOp))
            pl = _lib.OpList()
            pl.append(c)
            pl.time_ms(2)
            us = 1e3 * pl.time_ms(reps)
            fl, by = op_work(op, _lib)
            f.write(f"{j},{KIND_NAMES.get(op.kind, op.kind)},{op_dims(op, _lib)},{us:.2f},{fl / us / 1e6 if us else 0:.2f},{by / us / 1e3 if us else 0:.1f}\n")


def cpu_baseline(cfg, sd_cpu16, snap, frames=3, gpu_out=None):
    """The oracle (fp32 CPU restatement, `kind: port`) timed on the host cores on a bounded sample of the SAME
    workload, on the SAME inputs, weights, caches and ring-buffer state as the GPU leg (`snap`, taken right before
    the GPU frame whose output is `gpu_out`): the first, untimed oracle frame therefore is an end-to-end parity
    check of the full-size HIP frame against the oracle (`parity_rel_l2`); `frames` more frames are timed."""
    from oracle import unet_ref as O
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_update
    # The oracle is hundreds of small fp32 ops per frame: with one thread per logical CPU of the GPU box (256) every op
    # pays a 256-way fork/join and a frame took 355 s (round 1, profiles/r1f); 32 threads is where it stops scaling.
    ncpu = os.cpu_count() or 1
    threads = max(1, min(32, ncpu))
    torch.set_num_threads(threads)
    sd32 = {k: v.float() for k, v in sd_cpu16.items()}
    kv = [c.float() for c in snap["kv"]]
    rb = [t.clone() for t in snap["rb"]]
    x, d, enc, ts = snap["x"].float(), snap["d"].float(), snap["enc"].float(), snap["ts"]
    times, parity = [], None
    for f in range(frames + 1):
        t0 = time.perf_counter()
        ref = O.unet_forward(sd32, cfg, x, ts, enc, d, kv, temporal_attention_mask=rb[0], pe_idx=rb[1], update_idx=rb[2])
        times.append(time.perf_counter() - t0)
        if f == 0 and gpu_out is not None:
            a, b = gpu_out.double().cpu().reshape(-1), ref.double().reshape(-1)
            parity = {"rel_l2": float((a - b).norm() / b.norm()), "cosine": float(a @ b / (a.norm() * b.norm())),
                      "max_abs": float((a - b).abs().max()), "tolerance_rel_l2": 1e-2}
        ring_buffer_update(*rb, cfg.window_size, cfg.sink_size)
        if f >= 1 and sum(times) > 60.0:      # bounded sample: stop early on a slow host
            break
    nt = max(1, len(times) - 1)
    sec = sum(times[1:]) / nt if len(times) > 1 else times[0]
    more = None
    if ncpu >= 2 * threads and sec < 20.0:    # one frame at twice the threads: shows where the port stops scaling
        torch.set_num_threads(2 * threads)
        t0 = time.perf_counter()
        O.unet_forward(sd32, cfg, x, ts, enc, d, kv, temporal_attention_mask=rb[0], pe_idx=rb[1], update_idx=rb[2])
        more = {"threads": 2 * threads, "s_per_frame": round(time.perf_counter() - t0, 3)}
        torch.set_num_threads(threads)
    out = dict(value=1.0 / sec, unit="frames/s", cores=threads, kind="port",
               sample=f"{nt} timed frames (+1 untimed parity frame) of the same workload, inputs, caches and weights on the "
                      f"fp32 oracle, {sec:.2f} s/frame at {threads} threads; {ncpu} logical cpus on the box "
                      "(one thread per logical cpu measured 355 s/frame in round 1: fork/join bound)",
               s_per_frame_timed=[round(t, 3) for t in times[1:]], threads_x2=more)
    return out, parity


def baseline_metric() -> str:
    """BASELINE.json's metric string, verbatim (one streaming UNet step = one output frame)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:  # noqa: BLE001
        return "frames/sec @512\u00d7512, 2 denoise steps; per-step latency; 1/2/4/8 GPU"


def multi_stream_sweep(unet, cfg, args, dev, N, smax, frames=24):
    """Serving mode, informational: S independent frame streams on ONE GPU -- each a full UNet step of this workload with its
    own plan buffers, inputs and KV caches, all sharing the one packed-weight replica, each replayed from its own hipGraph on
    its own HIP stream.  A single stream is a chain of ~480 latency-bound launches that leaves most CUs idle most of the time;
    concurrent streams fill those gaps.  Reports ms per round (one frame of every stream = the per-stream frame latency) and
    aggregate frames/s for S = 1 .. smax.  `value` of the bench line stays the single-stream figure."""
    from live2diff_amd.unet_hip import HipStreamingUNet
    h, w, L = args.height // 8, args.width // 8, cfg.window_size
    g = torch.Generator(device=dev).manual_seed(1234)
    fns, keep = [], []
    for _ in range(smax):
        u = HipStreamingUNet(unet, cfg, h, w, N, device=dev, use_graph=True)       # shares unet's packed weights
        kvs = u.prepare_cache(N)
        for c in kvs:
            c.normal_(generator=g)
        x = torch.randn(N, 4, 1, h, w, device=dev, generator=g).half()
        dd = torch.randn(N, 4, 1, h, w, device=dev, generator=g).half()
        enc = torch.randn(N, 77, cfg.cross_attention_dim, device=dev, generator=g).half()
        ts = torch.tensor([399, 199, 99, 19][:N], device=dev)
        bias = torch.zeros(N, L, device=dev).half()
        pe = torch.arange(L, device=dev).repeat(N, 1)
        upd = torch.full((N,), L - 1, device=dev, dtype=torch.int64)
        keep.append((u, kvs))
        fns.append(lambda u=u, x=x, dd=dd, enc=enc, ts=ts, bias=bias, pe=pe, upd=upd, kvs=kvs: u(
            x, ts, encoder_hidden_states=enc, temporal_attention_mask=bias, depth_sample=dd, kv_cache=kvs, pe_idx=pe, update_idx=upd))
    streams = [torch.cuda.Stream(device=dev) for _ in range(smax)]
    for f in fns:                                   # first run direct (kernel attributes), second captures the graph
        f(); f(); f()
    torch.cuda.synchronize()
    out = {"note": "S independent streams per GPU (own KV caches / plan buffers, shared weights, one hipGraph and HIP stream each); "
                   "ms_per_round = one frame of every stream"}
    for S in range(1, smax + 1):
        def rnd():
            for f, st in zip(fns[:S], streams[:S]):
                with torch.cuda.stream(st):
                    f()
        for _ in range(4):
            rnd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            rnd()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / frames * 1e3
        out[f"S{S}"] = {"ms_per_round": round(ms, 3), "frames_per_s": round(S * 1e3 / ms, 2)}
    del fns, keep
    torch.cuda.empty_cache()
    return out


def pipeline_whole_frame(unet, cfg, args, dev, N, sink, kv):
    """The reference's published FPS definition (README.md:43-50 / test.py:201-205), measured through THIS repo's pipeline
    class: `StreamAnimateDiffusionDepth.__call__` per frame = preprocess + encode_image (TAESD encode, noise draw + add_noise)
    + encode_depth (384x384 resize, DPT-Hybrid detector, min-max / resize, TAESD encode) + predict_x0_batch (UNet, LCM step,
    shift register, re-noising, ring buffer: the device step) + decode_image().clip, with the reference's own timing tail
    (events around the call, a global synchronize, `inference_time_ema` = 0.9 ema + 0.1 t; pipeline :625-660).  Backends:
    HipStreamingUNet + HipTinyVAE + HipMidas + HipDepthGlue on random-init weights of the published architectures; `prepare()`
    runs the real warm-up (N warm-up UNet passes over 8 frames) first.  The text encoder runs once per prompt: excluded."""
    from types import SimpleNamespace

    from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict
    from live2diff_amd.pipeline_stream_animation_depth import StreamAnimateDiffusionDepth
    from live2diff_amd.vae_hip import HipTinyVAE, random_taesd_state_dict
    vae = HipTinyVAE(random_taesd_state_dict(device=dev), device=dev)
    detector = HipMidas(random_midas_state_dict(device=dev), device=dev)
    pipe = SimpleNamespace(device=dev, vae_scale_factor=8, unet=unet, vae=vae, depth_model=detector, scheduler=None)
    t_index = {1: [40], 2: [30, 40], 4: [25, 31, 37, 43]}.get(N, list(range(50 - 6 * N, 50, 6))[:N])
    s = StreamAnimateDiffusionDepth(pipe, num_inference_steps=50, t_index_list=t_index, width=args.width, height=args.height,
                                    do_add_noise=True, warmup_frames=sink, window_size=cfg.window_size)
    s.image_processor.assume_unit_range = True           # frames arrive in [0, 1] (what the reference's callers feed)
    gcpu = torch.Generator().manual_seed(4321)
    warm = [torch.rand(3, args.height, args.width, generator=gcpu) for _ in range(sink)]
    emb = torch.randn(1, 77, cfg.cross_attention_dim, generator=gcpu)
    s.prepare_cache(args.height, args.width, N)
    s.prepare(warm, prompt_embeds=emb, seed=3)
    s.enable_device_step()
    frames = [torch.rand(1, 3, args.height, args.width, generator=gcpu).to(dev) for _ in range(4)]
    nwarm, nw = 8, max(10, min(40, args.steps))
    for i in range(nwarm):
        out = s(frames[i % 4])
    s.inference_time_list.clear()
    torch.cuda.synchronize()
    tw = time.perf_counter()
    for i in range(nw):
        out = s(frames[i % 4])
    torch.cuda.synchronize()
    tw = (time.perf_counter() - tw) / nw
    lst = sorted(s.inference_time_list[-nw:])
    # opt-in pipelined mode (not in the reference): frame t + 1's encode / depth path on a second HIP stream under frame t's UNet step
    piped = None
    try:
        s.enable_frame_pipelining()
        s.push(frames[0])
        for i in range(nwarm):
            s.push(frames[(i + 1) % 4])
            out_p = s.pop()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(nw):
            s.push(frames[(i + 1) % 4])
            out_p = s.pop()
        torch.cuda.synchronize()
        tp = (time.perf_counter() - tp) / nw
        s.pop()
        piped = {"frames_per_s": round(1.0 / tp, 2), "ms_per_frame": round(1e3 * tp, 3), "finite": bool(torch.isfinite(out_p).all()),
                 "mode": "enable_frame_pipelining(): push(frame t+1) before pop(frame t); same outputs as __call__, one more frame in flight"}
    except Exception as e:  # noqa: BLE001
        piped = {"error": repr(e)}
    # the pipeline owned its own KV caches and conditioning: point the UNet's plan back at the benchmark's before they go away
    st = unet._plans["stream"]
    unet._bind_caches(st, kv)
    unet.invalidate_text_cache()
    return {"frames_per_s": round(1.0 / tw, 2), "ms_per_frame": round(1e3 * tw, 3),
            "inference_time_ema_ms": round(1e3 * s.inference_time_ema, 3), "inference_time_p50_ms": round(1e3 * lst[len(lst) // 2], 3),
            "depth_time_ema_ms": round(1e3 * s.depth_time_ema, 3), "finite": bool(torch.isfinite(out).all()), "pipelined": piped,
            "path": "StreamAnimateDiffusionDepth.__call__ (this repo's mirror of pipeline_stream_animation_depth.py:625-660) with "
                    "HipStreamingUNet + HipTinyVAE + HipMidas + device step, after prepare(); wall clock per call incl. the "
                    "reference's per-frame torch.cuda.synchronize()",
            "excludes": "text encoder (runs once per prompt)",
            "reference_published_context": "README.md:43-50: 16.43 FPS (TensorRT) / 6.91 FPS (none) on an RTX 4090, 512x512, 2 steps "
                                           "-- other hardware, context only",
            "vae_launches": {str(k): v_["n_ops"] for k, v_ in vae.plan_summary().items()},
            "depth_detector_launches": {str(k): v_["n_ops"] for k, v_ in detector.plan_summary().items()}}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` from a bare shell: re-exec this command line under torch.distributed.run, one rank per
    GPU of this node (rendezvous on 127.0.0.1, free port).  Rank 0 of the child job prints the JSON line."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs on this node, found {have}. Multi-GPU = one independent frame "
              f"stream per GPU (weak scaling); run with --gpus {max(have, 1)} or on a node with {n} GPUs.", file=sys.stderr)
        return 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL over xGMI between processes needs it here
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--graph", type=int, default=0, help="replay the plan from a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--tattn-variant", type=int, default=0)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--denoise-steps", type=int, default=2)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--sink", type=int, default=0, help="warm-up / sink slots (default 8; 4 with --window 12 = BASELINE configs[0])")
    ap.add_argument("--breakdown", type=int, default=1)
    ap.add_argument("--multi-stream", type=int, default=4, help="also measure 1..S independent streams on one GPU (informational; 0 = off)")
    ap.add_argument("--whole-frame", type=int, default=1, help="also time VAE encode x2 + depth detector and glue + UNet + VAE decode (informational)")
    ap.add_argument("--per-op", type=str, default="", help="write a per-launch timing CSV to this path")
    ap.add_argument("--dump-plan", type=str, default="", help="write the stream plan (one row per launch) as CSV")
    ap.add_argument("--device-step", type=int, default=0,
                    help="time the whole pipeline step on the device (UNet + LCM step + buffer shift + re-noising + ring "
                         "buffer as one plan, SURVEY 8f row F3) instead of the UNet boundary call; informational")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    from live2diff_amd import _lib, parallel
    from live2diff_amd.config import sd15_config
    from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init, ring_buffer_update
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import device_random_state_dict, random_state_dict, unet_param_spec

    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks. Launch with `python -m "
              f"torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`, or run `python "
              f"bench.py --gpus {args.gpus}` from a bare shell (it launches the ranks itself).", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sink = args.sink or (4 if args.window == 12 else 8)
    cfg = sd15_config(window_size=args.window, sink_size=sink)
    N = args.denoise_steps
    h, w = args.height // 8, args.width // 8
    default_workload = (args.height, args.width, N, args.window) == (512, 512, 2, 16)

    # ---- weights: generated on rank 0, one-time RCCL broadcast (the only collective of the data path)
    spec = unet_param_spec(cfg)
    need_cpu_weights = (world == 1 and not args.no_cpu_baseline)     # the CPU baseline runs the oracle on the same weights
    if rank == 0:
        sd_cpu = random_state_dict(cfg, dtype=torch.float16) if need_cpu_weights else device_random_state_dict(cfg, dev)
    else:
        sd_cpu = None
    if world > 1:
        # rank 0 packs ONCE and the PACKED weights are replicated (scatter + all-gather over all xGMI links, checksum);
        # the other ranks build their instance from what they received -- no packing pass, no raw state dict there
        unet0 = (HipStreamingUNet(sd_cpu, cfg, h, w, N, device=dev, use_graph=bool(args.graph), tattn_variant=args.tattn_variant)
                 if rank == 0 else None)
        packed = parallel.replicate_packed_weights(unet0, dev)
        unet = unet0 if rank == 0 else HipStreamingUNet(packed, cfg, h, w, N, device=dev, use_graph=bool(args.graph),
                                                        tattn_variant=args.tattn_variant)
        del packed
    else:
        sd = parallel.broadcast_state_dict(spec, sd_cpu, dev)
        unet = HipStreamingUNet(sd, cfg, h, w, N, device=dev, use_graph=bool(args.graph), tattn_variant=args.tattn_variant)
        del sd
    kv = unet.prepare_cache(N)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    for c in kv:
        c.normal_(generator=g)
    kv_bytes = sum(c.numel() * 2 for c in kv)
    x = torch.randn(N, 4, 1, h, w, generator=g, device=dev, dtype=torch.float16)
    d = torch.randn(N, 4, 1, h, w, generator=g, device=dev, dtype=torch.float16)
    enc = torch.randn(N, 77, cfg.cross_attention_dim, generator=g, device=dev, dtype=torch.float16)
    ts = torch.tensor([399, 199, 99, 19][:N], device=dev)
    rb = ring_buffer_init(N, cfg.window_size, cfg.sink_size)
    for _ in range(2 * cfg.window_size):          # steady state: every slot unmasked
        ring_buffer_update(*rb, cfg.window_size, cfg.sink_size)

    dstep = None
    if args.device_step:
        from live2diff_amd.scheduler import LCMSchedule
        from live2diff_amd.stream_step_hip import HipStreamStep
        sch = LCMSchedule()
        sch.set_timesteps(50)
        tl = ts.tolist()
        al = torch.tensor([float(sch.alphas_cumprod[t]) ** 0.5 for t in tl])
        be = torch.tensor([(1 - float(sch.alphas_cumprod[t])) ** 0.5 for t in tl])
        sc = [sch.get_scalings_for_boundary_condition_discrete(t) for t in tl]
        dstep = HipStreamStep(unet, kv, ts, enc, al, be, torch.tensor([float(c[0]) for c in sc]),
                              torch.tensor([float(c[1]) for c in sc]), use_graph=bool(args.graph))
        if N > 1:
            dstep.load_buffers(x[1:], d[1:])
        for _ in range(2 * cfg.window_size):      # same steady-state ring buffer, advanced on the device
            dstep.step(x[:1], d[:1])

    def step():
        if dstep is not None:
            return {"sample": dstep.step(x[:1], d[:1])}
        bias = rb[0].to(device=dev, dtype=torch.float16, non_blocking=True)
        pe_idx, upd = rb[1].to(dev, non_blocking=True), rb[2].to(dev, non_blocking=True)
        out = unet(x, ts, encoder_hidden_states=enc, temporal_attention_mask=bias, depth_sample=d, kv_cache=kv,
                   pe_idx=pe_idx, update_idx=upd)
        ring_buffer_update(*rb, cfg.window_size, cfg.sink_size)
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(mine, device=dev)
    per_rank_fps = [round(v, 3) for v in parallel.gather_floats(args.steps / mine, device=dev)]
    finite = bool(torch.isfinite(out["sample"]).all())
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed
    # per-step latency distribution (BASELINE metric: "...; per-step latency"): a separate pass AFTER the timed region,
    # events on the stream the plan is launched on, one per frame boundary; never allowed to fail the benchmark
    latency = None
    if rank == 0:
        try:
            n_lat = max(2, min(50, args.steps))
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_lat + 1)]
            evs[0].record()
            for i in range(n_lat):
                step()
                evs[i + 1].record()
            torch.cuda.synchronize()
            dts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n_lat))
            latency = {"p50_ms": round(dts[n_lat // 2], 4), "p95_ms": round(dts[min(n_lat - 1, int(0.95 * n_lat))], 4),
                       "max_ms": round(dts[-1], 4), "frames": n_lat}
        except Exception as e:  # noqa: BLE001
            latency = {"error": str(e)}

    # ---- whole frame (informational): what the published FPS figures include around the UNet --
    # TAESD encode of the frame, the depth path (384x384 resize, DPT-Hybrid depth detector, min-max + resize, TAESD encode of
    # the depth map), the UNet step, TAESD decode.  Random-init weights of the published architectures throughout.
    whole = None
    if rank == 0 and args.whole_frame and dstep is None and (args.height % 8 == 0 and args.width % 8 == 0):
        try:
            whole = pipeline_whole_frame(unet, cfg, args, dev, N, sink, kv)
        except Exception as e:  # noqa: BLE001  -- informational figure: never fails the benchmark
            whole = {"error": repr(e)}

    multi = None
    if rank == 0 and world == 1 and args.multi_stream > 1:
        try:
            multi = multi_stream_sweep(unet, cfg, args, dev, N, args.multi_stream)
        except Exception as e:  # noqa: BLE001  -- informational figure: never fails the benchmark
            multi = {"error": repr(e)}

    # CPU-baseline / parity leg, part 1 (before the per-kernel replays below, which run each kernel family out of context
    # and leave stale rows in the KV caches): same state for both legs = snapshot of (inputs, caches, ring buffer), then
    # ONE more GPU frame from exactly that state; the oracle starts from the snapshot after the JSON fields are assembled.
    snap = gpu_out = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        snap = {"kv": [c.cpu() for c in kv], "rb": [t.clone() for t in rb], "x": x.cpu(), "d": d.cpu(), "enc": enc.cpu(),
                "ts": ts.cpu()}
        if dstep is None:          # (the device step owns its ring state / noise: no like-for-like frame to compare)
            gpu_out = step()["sample"].float().cpu()
        torch.cuda.synchronize()

    result = {
        "metric": baseline_metric(),
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic", "latency_per_step": latency, "whole_frame": whole, "streams_per_gpu": multi,
        "per_rank_frames_per_s": per_rank_fps,
        "config": {"workload": f"{WORKLOAD_NAMES.get((args.height, args.width, N, args.window), 'custom (not a BASELINE config)')}: "
                               f"{args.height}x{args.width} image ({h}x{w} latent), {N} denoise steps, "
                               f"KV window L={cfg.window_size} ({sink} sink + {cfg.window_size - sink} rolling), SD-1.5 UNet widths + "
                               "Live2Diff temporal attention, one independent stream per GPU",
                   "params_M": 1277.7, "kv_cache_GB_per_stream": round(kv_bytes / 1e9, 3),
                   "plan_launches": unet.plan_summary()["n_ops"], "hipgraph": bool(args.graph),
                   "device": unet.device_name, "output_finite": finite, "default_workload": default_workload,
                   "timed_region": ("pipeline step on the device: UNet + LCM step + shift + noise + ring buffer (row F3)"
                                    if args.device_step else "UNet boundary call (+ host ring buffer and its upload)")},
    }
    my_frac = float("nan")
    if args.breakdown and rank != 0 and world > 1:
        # every rank replays its own kernel families: the N > 1 line carries each GPU's roofline fraction
        try:
            rows = per_kernel_breakdown(unet)
            name, r = max(rows.items(), key=lambda kv_: kv_[1]["ms"])
            if r["flops"] > 0 and name in MFMA_KERNELS:
                my_frac = r["flops"] / r["launches"] / (r["avg_us"] * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS
            else:
                my_frac = r["bytes"] / r["launches"] / (r["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
        except Exception:  # noqa: BLE001
            pass
    if rank == 0 and args.breakdown:
        rows = per_kernel_breakdown(unet)
        tot = sum(r["ms"] for r in rows.values())
        kernels = {}
        for name, r in sorted(rows.items(), key=lambda kv_: -kv_[1]["ms"]):
            ent = {"launches": r["launches"], "ms_per_frame": round(r["ms"], 4), "avg_us": round(r["avg_us"], 2)}
            if r["flops"] > 0:
                ent["tflops"] = round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2)
                ent["gbps"] = round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)
            kernels[name] = ent
        # matrix-pipe utilisation per family from the round's PMC passes (rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over the same
        # plan, tools/profile_round.sh step 4 -> profiles/mfma.json): static evidence, printed beside the live timings
        mpath = os.path.join(ROOT, "profiles", "mfma.json")
        mfma = json.load(open(mpath)).get("families", {}) if os.path.exists(mpath) else {}
        for name_, ent_ in kernels.items():
            u_ = mfma.get(name_) or mfma.get({"flash_attn_kernel": "flash_ring_kernel"}.get(name_, name_))
            if u_ and name_ in MFMA_KERNELS:
                ent_["mfma_util"] = u_["mfma_util"]
        result["kernels"] = kernels
        result["kernels_sum_ms"] = round(tot, 4)
        if args.window <= 16:
            result["tattn_variants_ms_per_frame"] = tattn_variant_sweep(unet)
        dom = max(rows.items(), key=lambda kv_: kv_[1]["ms"])
        name, r = dom
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))      # (the ring build of the flash op is traced under its own kernel name)
            traffic = (tj.get(name) or tj.get({"flash_attn_kernel": "flash_ring_kernel"}.get(name, name)) or {}).get("hbm_bytes_per_launch")
        tmeta = (tj.get("_meta") or {}) if os.path.exists(tpath) else {}
        tsrc = (f"static: profiles/traffic.json, collected in round {tmeta.get('round', '?')} with rocprofv3 FETCH_SIZE / WRITE_SIZE passes over "
                "the same plan (tools/traffic_frame.py), not inside this run") if traffic is not None else None
        if name in MFMA_KERNELS:
            ach = r["flops"] / r["launches"] / (r["avg_us"] * 1e-6) / 1e12
            result["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": tsrc,
                                  "traffic_round": tmeta.get("round"),
                                  "algorithmic_bytes_per_launch": round(r["bytes"] / r["launches"]),
                                  "traffic_over_algorithmic": (round(traffic / (r["bytes"] / r["launches"]), 2) if traffic else None),
                                  "mfma_util": kernels[name].get("mfma_util"),
                                  "note": "dominant kernel family by time; the figure comparable with the single igemm family of rounds 1-2 "
                                          "is roofline_gemm_kernels (igemm + rowgemm + pconv + wsgemm)"}
        else:
            ach = r["bytes"] / r["launches"] / (r["avg_us"] * 1e-6) / 1e9
            result["roofline"] = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": tsrc,
                                  "traffic_round": tmeta.get("round")}
        # The family times above are back-to-back REPLAYS of each family's launches; in the frame the same launches run between other
        # kernels (first touch of their code and operands, launch boundaries): the replay sum is below the measured step.  `frac` is the
        # in-frame figure -- the replay rate scaled by replay_sum / ms_per_step, i.e. every family charged its share of the frame's
        # remaining time (round-5 verdict: 0.156 in the rocprof trace vs 0.165 replayed) -- `frac_replay` the replayed one.
        scale = min(1.0, tot / result["ms_per_step"]) if result.get("ms_per_step") else 1.0
        result["roofline"]["frac_replay"] = result["roofline"]["frac"]
        result["roofline"]["achieved_replay"] = result["roofline"]["achieved"]
        result["roofline"]["frac"] = round(result["roofline"]["frac"] * scale, 4)
        result["roofline"]["achieved"] = round(result["roofline"]["achieved"] * scale, 2)
        result["roofline"]["in_frame_scale"] = round(scale, 4)
        my_frac = result["roofline"]["frac"]
        # Two families (wsgemm, igemm) have been within 1 % of each other since round 6: which one is "dominant" flips from run to run.
        # The other one is printed beside it, priced the same way, so that the line reads the same whichever way the tie falls.
        ranked = sorted(rows.items(), key=lambda kv_: -kv_[1]["ms"])
        if len(ranked) > 1 and ranked[1][1]["ms"] >= 0.9 * ranked[0][1]["ms"]:
            n2, r2 = ranked[1]
            if n2 in MFMA_KERNELS and r2["flops"] > 0:
                a2 = r2["flops"] / r2["launches"] / (r2["avg_us"] * 1e-6) / 1e12
                result["roofline"]["runner_up"] = {"kernel": n2, "bound": "mfma", "ms_per_frame": round(r2["ms"], 4), "achieved": round(a2 * scale, 2),
                                                   "unit": "TFLOP/s", "frac": round(a2 * scale / MFMA_PEAK_TFLOPS, 4)}
            else:
                a2 = r2["bytes"] / r2["launches"] / (r2["avg_us"] * 1e-6) / 1e9
                result["roofline"]["runner_up"] = {"kernel": n2, "bound": "hbm", "ms_per_frame": round(r2["ms"], 4), "achieved": round(a2 * scale, 1),
                                                   "unit": "GB/s", "frac": round(a2 * scale / HBM_PEAK_GBPS, 4)}
        # the frame's GEMM work is spread over three MFMA kernels (igemm / rowgemm / pconv) since round 3: their combined
        # rate is the figure comparable with the single igemm family of rounds 1-2
        GEMM_FAMILIES = ("igemm_kernel", "rowgemm_kernel", "pconv_kernel", "cconv_kernel", "wsgemm_kernel", "rowchain_kernel")
        gem = [rows[k_] for k_ in GEMM_FAMILIES if k_ in rows]
        if gem:
            gms, gfl = sum(g_["ms"] for g_ in gem), sum(g_["flops"] for g_ in gem)
            result["roofline_gemm_kernels"] = {"kernels": [k_ for k_ in GEMM_FAMILIES if k_ in rows],
                                               "bound": "mfma", "launches": sum(g_["launches"] for g_ in gem), "ms_per_frame": round(gms, 4),
                                               "achieved": round(gfl / (gms * 1e-3) / 1e12, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": round(gfl / (gms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
        # the decode-shaped half of the network (levels with M = N * T <= 512 tokens: weight-streaming problems, priced against the
        # HBM roofline): sum of algorithmic bytes of those GEMM launches / their time, replayed back to back (cold weights: every
        # launch has its own)
        try:
            import ctypes as _ct
            from live2diff_amd import _lib as _l
            st_ = unet._plans["stream"]
            small = []
            for j_ in range(len(st_.pl)):
                o_ = st_.pl[j_]
                m_ = {_l.OP_IGEMM: o_.i[13], _l.OP_ROWGEMM: o_.i[0], _l.OP_WSGEMM: o_.i[13], _l.OP_CCONV: o_.i[6] * o_.i[7] * o_.i[8]}.get(o_.kind)
                if m_ is not None and 0 < m_ <= 512:
                    small.append(o_)
            if small:
                pl_ = _l.OpList()
                for o_ in small:
                    c_ = _l.L2dOp()
                    _ct.memmove(_ct.byref(c_), _ct.byref(o_), _ct.sizeof(_l.L2dOp))
                    pl_.append(c_)
                pl_.time_ms(1)
                ms_ = pl_.time_ms(5)
                by_ = sum(op_work(o_, _l)[1] for o_ in small)
                fl_ = sum(op_work(o_, _l)[0] for o_ in small)
                result["roofline_small_m"] = {"what": "GEMM launches with M <= 512 tokens (levels 2 / 3 / mid at cfg-2)", "bound": "hbm",
                                              "launches": len(small), "ms_per_frame": round(ms_, 4), "algorithmic_bytes": round(by_),
                                              "achieved": round(by_ / (ms_ * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                              "frac": round(by_ / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                              "tflops": round(fl_ / (ms_ * 1e-3) / 1e12, 1),
                                              "kernels": sorted({KIND_NAMES.get(o_.kind, str(o_.kind)) for o_ in small})}
        except Exception as e_:          # noqa: BLE001  (informational leg)
            result["roofline_small_m"] = {"error": str(e_)[:200]}
        # the HBM-bound streaming KV-cache kernel is the one north_star singles out: always report it too
        t = rows.get("tattn_stream_kernel")
        if t:
            ach = t["bytes"] / (t["ms"] * 1e-3) / 1e9
            result["roofline_kv_cache_kernel"] = {"kernel": "tattn_stream_kernel", "bound": "hbm", "achieved": round(ach, 1),
                                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                                                  "traffic": ((json.load(open(tpath)).get("tattn_stream_kernel") or {}).get("hbm_bytes_per_launch")
                                                              if os.path.exists(tpath) else None)}
        # the flash kernel: priced against the MFMA peak; the floor evidence is the round-4 probe
        fl_ = rows.get("flash_attn_kernel")
        if fl_ and fl_["flops"] > 0:
            ach = fl_["flops"] / (fl_["ms"] * 1e-3) / 1e12
            result["roofline_flash"] = {"kernel": "flash_ring_kernel", "bound": "mfma", "launches": fl_["launches"],
                                        "ms_per_frame": round(fl_["ms"], 4), "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                                        "floor_evidence": "profiles/round4_n_exp_probe.txt: v_exp_f32 = 1.66 plain VALU issues per wave64 with two "
                                                          "waves per SIMD (2.62 alone); d = 40 loop body: 28 MFMA 16x16x32 (448 cycles) vs 59 VALU + 32 "
                                                          "exp (~448 cycles) per wave and 64-key tile, measured tile period 2344 cycles: bound by one "
                                                          "wave's softmax dependency chain at two waves per SIMD, not by either pipe"}
        # measured copy bandwidth for context
        try:
            import ctypes
            a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
            b = torch.empty_like(a)
            gb = ctypes.c_float(0)
            _lib.check(_lib.lib.l2d_copy_bench(a.data_ptr(), b.data_ptr(), a.numel() * 4, 5,
                                               ctypes.c_void_p(_lib.current_stream_ptr()), ctypes.byref(gb)), "copy_bench")
            result["hbm_copy_gbps_measured"] = round(float(gb.value), 1)
            # read-only streaming ceiling (the KV-cache kernel is 95 % reads): best of three launch geometries
            best = 0.0
            for unroll, bpc in ((1, 4), (4, 2), (2, 2)):
                _lib.check(_lib.lib.l2d_read_bench(a.data_ptr(), b.data_ptr(), a.numel() * 4, unroll, bpc, 5,
                                                   ctypes.c_void_p(_lib.current_stream_ptr()), ctypes.byref(gb)), "read_bench")
                best = max(best, float(gb.value))
            result["hbm_read_gbps_measured"] = round(best, 1)
            del a, b
        except Exception as e:  # noqa: BLE001
            result["hbm_copy_gbps_measured"] = f"error: {e}"
    if world > 1 and args.breakdown:
        result["per_rank_roofline_frac"] = [round(v, 4) for v in parallel.gather_floats(my_frac, device=dev)]
    if rank == 0 and args.per_op:
        per_op_table(unet, args.per_op)
    if rank == 0 and args.dump_plan:
        dump_plan(unet, args.dump_plan)
    if snap is not None:
        result["cpu_baseline"], result["parity_vs_oracle_full_size"] = cpu_baseline(cfg, sd_cpu, snap, frames=args.cpu_frames,
                                                                                   gpu_out=gpu_out)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
