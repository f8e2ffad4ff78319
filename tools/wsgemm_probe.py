"""Time the weight-streaming GEMM (wsgemm.hip) on the frame's small-token shapes against the kernel that serves them today
(igemm / rowgemm with their tuned schedules), weights COLD as in the frame.

    python tools/wsgemm_probe.py [--out gpurun_out/wsgemm_probe.json] [--quick]

In the frame every layer's weights are touched once per 9 ms, i.e. they come from HBM (the 256 MB Infinity Cache holds ~10 % of
the 2.56 GB).  A single op replayed in a loop would find its weights on-die, so each shape is timed as a plan of R launches over R
different weight copies (R x weight bytes >= 400 MB), same activations: the per-launch mean of `l2d_time_ops` is what a launch
costs with cold weights and L2-hot activations.  For wsgemm a grid of schedules (NW, NT, NL, S) is measured; the output of the
first copy is compared with the baseline kernel's.  Prints one table per shape and writes everything as JSON."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float16)


def shapes(quick):
    """(name, kind, dict) for the cfg-2 frame's M <= 512 GEMM launches (profiles/round3_final_frame_each_cfg2.csv)"""
    out = []
    for M in (512, 128):
        out += [
            (f"linear M{M} 1280->1280 (+res)", "lin", dict(M=M, K=1280, N=1280, res=True)),
            (f"LN+qkv M{M} 1280->3840", "lin", dict(M=M, K=1280, N=3840, pro=1)),
            (f"LN+GEGLU M{M} 1280->10240", "lin", dict(M=M, K=1280, N=10240, pro=1, epi=1)),
            (f"FF2 M{M} 5120->1280 (+res)", "lin", dict(M=M, K=5120, N=1280, res=True)),
            (f"conv3x3 M{M} 1280->1280", "conv", dict(B=2, H=(16 if M == 512 else 8), C1=1280, C2=0, N=1280)),
        ]
        if not quick:
            out += [
                (f"conv3x3 M{M} 2560->1280 (concat)", "conv", dict(B=2, H=(16 if M == 512 else 8), C1=1280, C2=1280, N=1280)),
                (f"shortcut M{M} 2560->1280 (concat)", "lin", dict(M=M, K=1280, K2=1280, N=1280)),
            ]
    if not quick:
        out += [("LN+GEGLU M2048 640->5120", "lin", dict(M=2048, K=640, N=5120, pro=1, epi=1)),
                ("linear M2048 640->640 (+res)", "lin", dict(M=2048, K=640, N=640, res=True)),
                ("FF2 M2048 2560->640 (+res)", "lin", dict(M=2048, K=2560, N=640, res=True))]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "wsgemm_probe.json"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--cold-mb", type=int, default=400)
    ap.add_argument("--dry", action="store_true", help="CPU: build every op and push it through the library's argument validation only")
    args = ap.parse_args()
    from live2diff_amd import _lib, ops as L
    global DEV
    if args.dry:
        DEV = "cpu"
        _lib.set_dry_run(True)
        args.cold_mb = 1
    else:
        print("device:", _lib.device_name())
    results = []
    cnt = torch.zeros(1 << 16, dtype=torch.int32, device=DEV)

    def time_plan(build, R):
        pl = _lib.OpList()
        for r in range(R):
            op, keep = build(r)
            pl.append(op, *keep)
        pl.run()
        if args.dry:
            return 1.0
        torch.cuda.synchronize()
        pl.time_ms(1)
        return pl.time_ms(args.reps) * 1000.0 / R

    for name, kind, d in shapes(args.quick):
        torch.manual_seed(0)
        if kind == "lin":
            M, K, N = d["M"], d["K"], d["N"]
            K2 = d.get("K2", 0)
            Kt = K + K2
            pro, epi = d.get("pro", 0), d.get("epi", 0)
            wbytes = N * Kt * 2
            R = max(2, min(96, -(-args.cold_mb * (1 << 20) // wbytes)))
            x = rnd(M, K, seed=1).to(DEV)
            x2 = rnd(M, K2, seed=5).to(DEV) if K2 else None
            No = N // 2 if epi == 1 else N
            res = rnd(M, No, seed=2).to(DEV) if d.get("res") else None
            w = rnd(N, Kt, seed=3, scale=Kt ** -0.5).to(DEV)
            b = rnd(N, seed=4).float().to(DEV)
            gm = (1 + 0.1 * rnd(Kt, seed=6).float()).half().to(DEV) if pro else None
            bt = (0.1 * rnd(Kt, seed=7).float()).half().to(DEV) if pro else None
            wp, bp, cs = L.pack_wsgemm(w, b, gm, bt, geglu=(epi == 1))
            wps = [wp] + [wp.clone() for _ in range(R - 1)]
            out_ws = torch.zeros(M, No, dtype=torch.float16, device=DEV)
            out_base = torch.zeros_like(out_ws)
            # ---- baseline: what the round-3 plan launches for this layer
            if pro or (K2 == 0 and Kt <= 640):
                wr, br = L.pack_rowgemm(w, b, gm, bt, geglu=(epi == 1))
                wrs = [wr] + [wr.clone() for _ in range(R - 1)]
                base = lambda r: L.rowgemm(x, wrs[r], out_base, M=M, K=K, Nout=N, ldx=K, ldo=No, bias=br, res=res, ldr=(No if res is not None else 0),
                                           epi=epi, pro=pro, eps=1e-5, T=M // 2)
                base_name = "rowgemm"
            else:
                wi = L.pack_linear(w)
                wis = [wi] + [wi.clone() for _ in range(R - 1)]
                tile, S0, variant = L.igemm_schedule(M, N, Kt, 1, 0, 1)
                kw = {}
                if S0 > 1:
                    n_ws, n_cnt = L.splitk_sizes(M, N, S0, 1, tile)
                    kw = dict(ws=torch.empty(n_ws, dtype=torch.float32, device=DEV), cnt=cnt, cnt_off=0)
                base = lambda r: L.igemm(x, wis[r], out_base, M=M, Nout=N, C1=K, ldx1=K, CinP=Kt, ldo=N, x2=x2, C2=K2, ldx2=K2, bias=b, res=res,
                                         ldr=(N if res is not None else 0), splitk=S0, tile=tile, variant=variant, order=1, **kw)
                base_name = f"igemm t{tile} S{S0} v{variant}"

            def ws_build(sched):
                NW, NT, NL, S, ntw = sched
                kw = {}
                if S > 1:
                    n_ws, n_cnt = L.wsgemm_sizes(M, N, NW, NT, S)
                    kw = dict(ws=torch.empty(n_ws, dtype=torch.float32, device=DEV), cnt=cnt, cnt_off=1024)
                return lambda r: L.wsgemm(x, wps[r], out_ws, M=M, Nout=N, C1=K, ldx1=K, x2=x2, C2=K2, ldx2=K2, ldo=No, bias=bp, colsum=cs, res=res,
                                          ldr=(No if res is not None else 0), epi=epi, pro=pro, eps=1e-5, T=M // 2, sched=sched, **kw)
            Ktot, ntr, taps = Kt, 0, 1
        else:
            B, H, C1, C2, N = d["B"], d["H"], d["C1"], d["C2"], d["N"]
            M, Cin = B * H * H, C1 + C2
            Ktot, pro, epi, ntr, taps = 9 * Cin, 0, 0, 0, 9
            wbytes = N * Ktot * 2
            R = max(2, min(96, -(-args.cold_mb * (1 << 20) // wbytes)))
            x = rnd(M, C1, seed=1).to(DEV)
            x2 = rnd(M, C2, seed=5).to(DEV) if C2 else None
            res = rnd(M, N, seed=2).to(DEV)
            w = rnd(N, Cin, 3, 3, seed=3, scale=Ktot ** -0.5).to(DEV)
            b = rnd(N, seed=4).float().to(DEV)
            temb = rnd(B, N, seed=8).float().to(DEV)
            wp = L.pack_wsgemm_conv3x3(w)
            wps = [wp] + [wp.clone() for _ in range(R - 1)]
            wi = L.pack_conv3x3(w)
            wis = [wi] + [wi.clone() for _ in range(R - 1)]
            out_ws = torch.zeros(M, N, dtype=torch.float16, device=DEV)
            out_base = torch.zeros_like(out_ws)
            tile, S0, variant = L.igemm_schedule(M, N, Ktot, 1, 0, 9)
            kwb = {}
            if S0 > 1:
                if L.splitk_fused(S0):
                    n_ws, n_cnt = L.splitk_sizes(M, N, S0, 1, tile)
                    kwb = dict(ws=torch.empty(n_ws, dtype=torch.float32, device=DEV), cnt=cnt, cnt_off=0)
                else:
                    kwb = dict(ws=torch.empty(S0 * M * N, dtype=torch.float32, device=DEV))
            base = lambda r: L.igemm(x, wis[r], out_base, M=M, Nout=N, C1=C1, ldx1=C1, CinP=Cin, ldo=N, x2=x2, C2=C2, ldx2=C2, bias=b, rowbias=temb,
                                     ldrb=N, rows_per_bias=H * H, res=res, ldr=N, taps=9, B=B, Hin=H, Win=H, Hout=H, Wout=H, splitk=S0, tile=tile,
                                     variant=variant, order=1, **kwb)
            base_name = f"igemm t{tile} S{S0} v{variant}"

            def ws_build(sched):
                NW, NT, NL, S, ntw = sched
                kw = {}
                if S > 1:
                    n_ws, n_cnt = L.wsgemm_sizes(M, N, NW, NT, S)
                    kw = dict(ws=torch.empty(n_ws, dtype=torch.float32, device=DEV), cnt=cnt, cnt_off=1024)
                return lambda r: L.wsgemm(x, wps[r], out_ws, M=M, Nout=N, C1=C1, ldx1=C1, x2=x2, C2=C2, ldx2=C2, ldo=N, bias=b, rowbias=temb, ldrb=N,
                                          rows_per_bias=H * H, res=res, ldr=N, taps=9, B=B, H=H, W=H, sched=sched, **kw)

        t_base = time_plan(base, R)
        nm = -(-M // 128)
        nch = Ktot // 64
        tiles = N // 32
        rows = []
        for nw, nt in ((1, 1), (2, 1), (4, 1), (8, 1), (5, 1)):
            if tiles % (nw * nt):
                continue
            ny = tiles // (nw * nt)
            for S in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 30):
                blocks = nm * ny * S
                if S > max(1, nch // 2) or blocks > 1280 or (blocks < 40 and S < 8) or nch / S < 2:
                    continue
                for nl in (1, 2):
                    sched = (nw, nt, nl, S, nm == 1)
                    try:
                        t = time_plan(ws_build(sched), R)
                    except Exception as e:          # noqa: BLE001
                        rows.append(dict(sched=sched, error=str(e)[:200]))
                        continue
                    err = ((out_ws.double() - out_base.double()).norm() / out_base.double().norm().clamp_min(1e-9)).item()
                    rows.append(dict(sched=sched, us=round(t, 2), blocks=blocks, relerr_vs_base=err))
        good = sorted([r for r in rows if "us" in r], key=lambda r: r["us"])
        print(f"\n== {name}: baseline {base_name} {t_base:.1f} us; weights {wbytes / 1e6:.1f} MB x R={R}; baseline {wbytes / t_base / 1e6:.2f} TB/s")
        for r in good[:8]:
            print(f"   wsgemm {str(r['sched']):28s} {r['us']:7.1f} us  blocks {r['blocks']:4d}  {wbytes / r['us'] / 1e6:5.2f} TB/s  err {r['relerr_vs_base']:.1e}")
        bad = [r for r in rows if "error" in r or r.get("relerr_vs_base", 0) > 3e-3]
        for r in bad[:5]:
            print("   !!", r)
        results.append(dict(name=name, baseline=base_name, baseline_us=round(t_base, 2), weight_bytes=wbytes, R=R, rows=rows))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
