"""Find the first launch of the tiny UNet plan whose result depends on what else runs on the GPU: every op is run on its own
(synchronised), all arena buffers hashed after it; once with an idle GPU, once with the depth detector looping on a second stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from live2diff_amd import _lib
from live2diff_amd.config import tiny_config
from live2diff_amd.unet_hip import HipStreamingUNet
from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict
from live2diff_amd.weights import random_state_dict
import bench
DEV = "cuda"
cfg = tiny_config(channels=(64, 128, 128, 128), cross_attention_dim=64)
sd = {k: v.to(DEV) for k, v in random_state_dict(cfg, dtype=torch.float16).items()}
unet = HipStreamingUNet(sd, cfg, 16, 16, 2)
kv = unet.prepare_cache(2)
g = torch.Generator().manual_seed(1)
for c in kv:
    c.copy_(torch.randn(c.shape, generator=g).half())
kv0 = [c.clone() for c in kv]
st = unet._plan("stream", kv)
st.in_sample.copy_(torch.randn(st.in_sample.shape, generator=g).half()); st.in_depth.copy_(torch.randn(st.in_depth.shape, generator=g).half())
st.in_enc.copy_(torch.randn(st.in_enc.shape, generator=g).half()); st.in_t.copy_(torch.tensor([399, 199]))
st.in_pe_idx.copy_(torch.arange(cfg.window_size).repeat(2, 1)); st.in_upd.copy_(torch.tensor([3, 5]))
st.cond_pl.run(); torch.cuda.synchronize()
det = HipMidas(random_midas_state_dict(), device=DEV)
img = torch.rand(1, 3, 384, 384, device=DEV).half()
side = torch.cuda.Stream()

def sig():
    tot = 0
    for t in st.arena.all:
        tot = (tot * 1000003 + int(t.view(torch.uint8).long().sum().item())) % (1 << 61)
    tot = (tot * 1000003 + int(st.out_sample.view(torch.uint8).long().sum().item())) % (1 << 61)
    return tot

def run(busy):
    for c, c0 in zip(kv, kv0):
        c.copy_(c0)
    for t in st.arena.all:
        t.zero_()
    st.out_sample.zero_()
    torch.cuda.synchronize()
    out = []
    for j in range(len(st.pl)):
        pl = _lib.OpList(); c = _lib.L2dOp(); ctypes.memmove(ctypes.byref(c), ctypes.byref(st.pl[j]), ctypes.sizeof(_lib.L2dOp)); pl.append(c)
        if busy:
            with torch.cuda.stream(side):
                det(img)
        pl.run()
        torch.cuda.synchronize()
        out.append(sig())
    return out

a = run(False); a2 = run(False)
print("solo vs solo identical:", a == a2)
for rep in range(4):
    b = run(True)
    bad = [j for j in range(len(a)) if a[j] != b[j]]
    if bad:
        j = bad[0]
        op = st.pl[j]
        print(f"rep {rep}: first differing op #{j} kind {bench.KIND_NAMES.get(op.kind, op.kind)} {bench.op_dims(op, _lib)}; {len(bad)} ops differ after it")
    else:
        print(f"rep {rep}: no difference")
