"""ms per depth-detector forward (HipMidas plan replay, HIP events) + per-kernel-family split via l2d_time_ops on sub-plans."""
import json
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from live2diff_amd import _lib
from live2diff_amd.midas_hip import HipMidas, random_midas_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = HipMidas(random_midas_state_dict(device="cuda"), device="cuda")
x = torch.randn(B, 3, 384, 384, device="cuda").half()
m(x)
torch.cuda.synchronize()
st = m._plans[(B, 384, 384)]
ms = st.pl.time_ms(reps=20)
fam = {}
names = {v: k for k, v in vars(_lib).items() if k.startswith("OP_")}
for op in st.pl._ops:
    one = _lib.OpList([op])
    fam.setdefault(names.get(op.kind, str(op.kind)), []).append(one.time_ms(reps=5))
print(json.dumps({"B": B, "ms_per_forward": ms, "n_ops": len(st.pl), "arena_MB": st.arena_bytes / 2**20,
                  "families_ms": {k: [len(v), round(sum(v), 3)] for k, v in sorted(fam.items(), key=lambda kv: -sum(kv[1]))}}))
