"""Probe: does running the N denoise rows of the stream batch as N concurrent single-row plans (one HIP stream each) hide the
per-launch floor?  Times (a) one N=2 UNet, (b) two N=1 UNets launched back to back on ONE stream, (c) the same two on TWO
streams.  Same weights (shared tensors), private KV caches.  Output: one JSON line."""
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
graph = int(os.environ.get("PROBE_GRAPH", "1"))
L = cfg.window_size


def make(N):
    u = HipStreamingUNet(sd, cfg, 64, 64, N, device=dev, use_graph=bool(graph))
    kv = u.prepare_cache(N)
    for c in kv:
        c.normal_()
    x = torch.randn(N, 4, 1, 64, 64, device=dev).half()
    enc = torch.randn(N, 77, cfg.cross_attention_dim, device=dev).half()
    ts = torch.tensor([399, 199][:N], device=dev)
    bias = torch.zeros(N, L, device=dev).half()
    pe = torch.arange(L, device=dev).repeat(N, 1)
    upd = torch.full((N,), 9, device=dev, dtype=torch.int64)
    return lambda: u(x, ts, encoder_hidden_states=enc, temporal_attention_mask=bias, depth_sample=x, kv_cache=kv, pe_idx=pe, update_idx=upd)


def timeit(fn, n=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


both = make(2)
a, b = make(1), make(1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
res = {"graph": graph, "n2_ms": timeit(both), "n1_ms": timeit(a)}


def serial():
    a(); b()


def parallel():
    with torch.cuda.stream(s1):
        a()
    with torch.cuda.stream(s2):
        b()


res["two_n1_one_stream_ms"] = timeit(serial)
res["two_n1_two_streams_ms"] = timeit(parallel)
print(json.dumps(res))
