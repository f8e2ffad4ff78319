"""Time the streaming temporal-attention launches of a configuration (40 launches, KV caches resident), HIP events:
    L2D_TATTN_RING=<geo> python tools/tattn_time.py [--height 512 --width 512 --denoise-steps 2 --window 16]
Prints ms per frame for the 40 launches and the algorithmic GB/s (SURVEY 8d bytes)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--denoise-steps", type=int, default=2)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--variant", type=int, default=0, help="0 auto (ring where it applies), 2 chunked kernel, 13 ring forced")
    a = ap.parse_args()
    from live2diff_amd import _lib, ops
    from live2diff_amd.config import motion_module_layout, sd15_config
    dev = "cuda"
    cfg = sd15_config(window_size=a.window, sink_size=(4 if a.window == 12 else 8))
    N, L = a.denoise_steps, a.window
    pl = _lib.OpList()
    g = torch.Generator(device=dev).manual_seed(0)
    byts = 0
    pe_idx = torch.arange(L, device=dev).repeat(N, 1).contiguous()
    upd = torch.full((N,), L - 1, dtype=torch.int64, device=dev)
    bias = torch.zeros(N, L, dtype=torch.float16, device=dev)
    for (C, hh, ww, _l) in motion_module_layout(cfg, a.height // 8, a.width // 8):
        T = hh * ww
        cache = torch.randn(N, 2, T, L, C, device=dev, generator=g, dtype=torch.float16)
        qkv = torch.randn(N * T, 3 * C, device=dev, generator=g, dtype=torch.float16)
        pe = [torch.randn(L, C, device=dev, generator=g, dtype=torch.float16) for _ in range(3)]
        out = torch.empty(N * T, C, device=dev, dtype=torch.float16)
        pl.append(*ops.tattn_stream(qkv, cache, pe[0], pe[1], pe[2], pe_idx, upd, bias, out, N=N, T=T, C=C, L=L, H=8, variant=a.variant))
        byts += 4 * N * T * L * C + 8 * N * T * C
    pl.run()
    torch.cuda.synchronize()
    ms = min(pl.time_ms(5) for _ in range(3))
    print(json.dumps({"L2D_TATTN_RING": os.environ.get("L2D_TATTN_RING", ""), "window": L, "variant": a.variant, "launches": len(pl), "ms_per_frame": round(ms, 4),
                      "GBps": round(byts / ms / 1e6, 1), "frac_of_8TBps": round(byts / ms / 1e6 / 8000, 4)}))


if __name__ == "__main__":
    main()
