"""Analysis tool (not product): bit-stability of the full cfg-2 UNet frame under concurrent load.  The frame's op list is replayed REPS
times from the same inputs and KV caches while the depth detector loops on a second stream; every replay's output and caches are
compared bit for bit with an idle-GPU replay.  L2D_WSGEMM=0 in the environment restricts the frame to the round-3 kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from live2diff_amd.config import sd15_config
from live2diff_amd.unet_hip import HipStreamingUNet
from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict
from live2diff_amd.weights import device_random_state_dict
DEV = torch.device("cuda", 0)
REPS = int(os.environ.get("REPS", "300"))
H = int(os.environ.get("HW", "64")); N = int(os.environ.get("NSTEP", "2"))
cfg = sd15_config(window_size=16, sink_size=8)
sd = device_random_state_dict(cfg, DEV)
unet = HipStreamingUNet(sd, cfg, H, H, N, device=DEV)
del sd
kv = unet.prepare_cache(N)
g = torch.Generator(device=DEV).manual_seed(7)
for c in kv:
    c.normal_(generator=g)
kv0 = [c.clone() for c in kv]
st = unet._plan("stream", kv)
for t, shape in ((st.in_sample, None), (st.in_depth, None), (st.in_enc, None)):
    t.copy_(torch.randn(t.shape, generator=g, device=DEV, dtype=torch.float16))
st.in_t.copy_(torch.tensor([399, 199, 99, 19][:N])); st.in_pe_idx.copy_(torch.arange(cfg.window_size).repeat(N, 1)); st.in_upd.copy_(torch.tensor([3, 5, 7, 9][:N]))
st.cond_pl.run(); torch.cuda.synchronize()
det = HipMidas(random_midas_state_dict(), device=DEV)
img = torch.rand(1, 3, 384, 384, device=DEV).half()
side = torch.cuda.Stream()

def frame(busy):
    for c, c0 in zip(kv, kv0):
        c.copy_(c0)
    torch.cuda.synchronize()
    if busy:
        with torch.cuda.stream(side):
            for _ in range(4):
                det(img)
    st.pl.run()
    torch.cuda.synchronize()
    return [st.out_sample.clone()] + [c.clone() for c in kv]

ref = frame(False)
same = all(torch.equal(a, b) for a, b in zip(ref, frame(False)))
print("idle vs idle identical:", same, "| ops per frame:", len(st.pl), "| L2D_WSGEMM =", os.environ.get("L2D_WSGEMM", "default"))
bad = 0
for rep in range(REPS):
    out = frame(True)
    if not all(torch.equal(a, b) for a, b in zip(ref, out)):
        bad += 1
        if bad <= 5:
            d = (out[0].float() - ref[0].float()).abs()
            print(f"  rep {rep}: output differs: max {d.max().item():.4g} at {int((d > 0).sum())} elements; caches differing: "
                  f"{sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref[1:], out[1:]))}")
print(f"frames differing from the idle-GPU frame: {bad} / {REPS}", flush=True)
