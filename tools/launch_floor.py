"""Analysis tool: the per-launch floor of the plan executor on this machine -- N back-to-back launches of the smallest ops the library
has (16-byte device copy = hipMemcpyAsync node; a 1-row timestep embedding = one 64-thread kernel; a 64-token x 64-channel GroupNorm
apply = one small block), direct launches and hipGraph replay.  What a frame of ~480 launches pays before any kernel does work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from live2diff_amd import _lib, ops
DEV = "cuda"
N = int(os.environ.get("NLAUNCH", "483"))
a, b = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
t = torch.tensor([399], device=DEV)
emb = torch.zeros(1, 320, dtype=torch.float16, device=DEV)
x = torch.randn(64, 64, device=DEV).half(); y = torch.empty_like(x)
part = torch.zeros(1 * 1 * 32 * 2, device=DEV)
gam, bet = torch.ones(64, device=DEV).half(), torch.zeros(64, device=DEV).half()
kw = dict(B=1, T=64, C1=64, ld1=64, G=32, nchunk=1)
cases = {
    "copy 16 B": lambda: ops.copy(a, b, 16),
    "timestep_embed (1 block of 256 threads)": lambda: ops.timestep_embed(t, emb, N=1, dim=320),
    "gn_apply 64 x 64 (dependent chain: every launch reads what the previous wrote)": None,
}
for name, mk in cases.items():
    pl = _lib.OpList()
    if mk is None:
        ops.run(ops.gn_stats(x, part, **kw))
        src, dst = x, y
        for _ in range(N):
            pl.append(*ops.gn_apply(src, part, gam, bet, dst, eps=1e-5, silu=False, **kw))
            src, dst = dst, src
    else:
        for _ in range(N):
            pl.append(*mk())
    pl.run(); torch.cuda.synchronize()
    ms = pl.time_ms(10)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # (capture is not allowed on the legacy default stream)
        g = _lib.Graph(pl, stream=int(side.cuda_stream))
        g.launch(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.launch()
        e1.record(); torch.cuda.synchronize()
    print(f"{name}: {N} launches: direct {ms:.3f} ms = {1e3 * ms / N:.2f} us per launch; hipGraph {e0.elapsed_time(e1) / 10:.3f} ms = {1e2 * e0.elapsed_time(e1) / N:.2f} us per launch", flush=True)
