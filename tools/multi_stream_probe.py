"""Probe: S independent frame streams (each the full cfg-2 UNet step, N = 2 denoise rows, private KV caches and plan buffers,
replicated packed weights) on S HIP streams of ONE GPU.  The per-stream step is a chain of ~480 latency-bound launches; do
concurrent streams fill each other's launch gaps and idle CUs?  Output: one JSON line with ms per round (one frame of every
stream) and aggregate frames/s for S = 1 .. 4, direct launches and hipGraph replay."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd.config import sd15_config                                    # noqa: E402
from live2diff_amd.unet_hip import HipStreamingUNet                             # noqa: E402
from live2diff_amd.weights import device_random_state_dict                      # noqa: E402

dev = torch.device("cuda")
cfg = sd15_config(window_size=16, sink_size=8)
sd = device_random_state_dict(cfg, dev)
L, N = cfg.window_size, 2
SMAX = int(os.environ.get("PROBE_STREAMS", "4"))


def make(graph):
    u = HipStreamingUNet(sd, cfg, 64, 64, N, device=dev, use_graph=bool(graph))
    kv = u.prepare_cache(N)
    for c in kv:
        c.normal_()
    x = torch.randn(N, 4, 1, 64, 64, device=dev).half()
    enc = torch.randn(N, 77, cfg.cross_attention_dim, device=dev).half()
    ts = torch.tensor([399, 199], device=dev)
    bias = torch.zeros(N, L, device=dev).half()
    pe = torch.arange(L, device=dev).repeat(N, 1)
    upd = torch.full((N,), 9, device=dev, dtype=torch.int64)
    return lambda: u(x, ts, encoder_hidden_states=enc, temporal_attention_mask=bias, depth_sample=x, kv_cache=kv, pe_idx=pe, update_idx=upd)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


res = {}
for graph in (0, 1):
    fns = [make(graph) for _ in range(SMAX)]
    streams = [torch.cuda.Stream() for _ in range(SMAX)]
    for f in fns:                                   # first run of every plan is direct (LDS attributes, graph capture)
        f(); f()
    torch.cuda.synchronize()
    for S in range(1, SMAX + 1):
        def rnd(S=S):
            for f, st in zip(fns[:S], streams[:S]):
                with torch.cuda.stream(st):
                    f()
        ms = timeit(rnd)
        res[f"graph{graph}_S{S}"] = {"ms_per_round": round(ms, 3), "frames_per_s": round(S * 1e3 / ms, 1)}
    del fns
    torch.cuda.empty_cache()
print(json.dumps(res))
