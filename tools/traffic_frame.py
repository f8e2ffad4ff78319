"""Target of the HBM-traffic PMC passes (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python tools/traffic_frame.py):
builds the cfg-2 UNet plan and replays the frame FRAMES times, nothing else (no timing events, no CPU baseline, no second stream --
round 4's passes over the whole bench.py died inside rocprofv3).  Every launch is the in-frame one: real weights (cold: 2.56 GB of
them pass between two uses), real neighbours."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from live2diff_amd.config import sd15_config
from live2diff_amd.unet_hip import HipStreamingUNet
from live2diff_amd.weights import device_random_state_dict
DEV = torch.device("cuda", 0)
FRAMES = int(os.environ.get("FRAMES", "3"))
H, W, N, L = (int(os.environ.get(k, d)) for k, d in (("LAT_H", "64"), ("LAT_W", "64"), ("NSTEP", "2"), ("WINDOW", "16")))
cfg = sd15_config(window_size=L, sink_size=8)
unet = HipStreamingUNet(device_random_state_dict(cfg, DEV), cfg, H, W, N, device=DEV)
kv = unet.prepare_cache(N)
g = torch.Generator(device=DEV).manual_seed(3)
for c in kv:
    c.normal_(generator=g)
st = unet._plan("stream", kv)
for t in (st.in_sample, st.in_depth, st.in_enc):
    t.copy_(torch.randn(t.shape, generator=g, device=DEV, dtype=torch.float16))
st.in_t.copy_(torch.tensor([399, 199, 99, 19][:N])); st.in_pe_idx.copy_(torch.arange(L).repeat(N, 1)); st.in_upd.copy_(torch.tensor([9, 11, 13, 15][:N]))
st.cond_pl.run()
for _ in range(FRAMES):
    st.pl.run()
torch.cuda.synchronize()
print("frames", FRAMES, "ops", len(st.pl), flush=True)
