"""Upper bound for cross-launch weight prefetch: every launch of the cfg-2 frame timed IN the frame (l2d_time_each: cold weights,
real neighbours) and ALONE in a loop (its weights on-die after the first pass).  Prints per kernel family and per token count
the two sums, and the launches that gain most.   python tools/warm_vs_cold.py [--out gpurun_out/warm_vs_cold.json]"""
import argparse
import collections
import ctypes
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "warm_vs_cold.json"))
    args = ap.parse_args()
    from live2diff_amd import _lib
    from live2diff_amd.config import sd15_config
    from live2diff_amd.unet_hip import HipStreamingUNet
    from live2diff_amd.weights import device_random_state_dict
    import bench
    dev = torch.device("cuda", 0)
    cfg = sd15_config(window_size=16, sink_size=8)
    unet = HipStreamingUNet(device_random_state_dict(cfg, dev), cfg, 64, 64, 2, device=dev)
    kv = unet.prepare_cache(2)
    for c in kv:
        c.normal_()
    st = unet._plan("stream", kv)
    st.cond_pl.run()
    st.pl.run()
    torch.cuda.synchronize()
    st.pl.time_each_us(1)
    cold = st.pl.time_each_us(8)
    rows = []
    for j in range(len(st.pl)):
        op = st.pl[j]
        c = _lib.L2dOp()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(op), ctypes.sizeof(_lib.L2dOp))
        pl = _lib.OpList()
        pl.append(c)
        pl.time_ms(2)
        warm = 1e3 * pl.time_ms(10)
        fl, by = bench.op_work(op, _lib)
        dims = bench.op_dims(op, _lib)
        m = re.search(r"\bM(\d+)", dims)
        rows.append(dict(idx=j, kernel=bench.KIND_NAMES.get(op.kind, str(op.kind)), dims=dims, cold_us=cold[j], warm_us=warm, bytes=by,
                         M=int(m.group(1)) if m else 0))
    fam = collections.OrderedDict()
    for r in rows:
        k = r["kernel"] + (f" M{r['M']}" if r["M"] and r["kernel"] in ("igemm_kernel", "rowgemm_kernel", "wsgemm_kernel") else "")
        a = fam.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += r["cold_us"]; a[2] += r["warm_us"]
    print(f"frame: in-frame sum {sum(r['cold_us'] for r in rows) / 1e3:.3f} ms, alone (warm, incl. launch floor) {sum(r['warm_us'] for r in rows) / 1e3:.3f} ms")
    for k, a in sorted(fam.items(), key=lambda kv_: -(kv_[1][1] - kv_[1][2])):
        print(f"  {k:28s} n={a[0]:3d}  in-frame {a[1] / 1e3:6.3f} ms   alone {a[2] / 1e3:6.3f} ms   delta {(a[1] - a[2]) / 1e3:6.3f}")
    print("top single launches by delta:")
    for r in sorted(rows, key=lambda r: -(r["cold_us"] - r["warm_us"]))[:25]:
        print(f"  #{r['idx']:3d} {r['kernel']:16s} {r['dims'][:70]:70s} in-frame {r['cold_us']:6.1f} alone {r['warm_us']:6.1f}")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"))


if __name__ == "__main__":
    main()
