"""Analysis tool (not product): localises a violation of the stream-batch row symmetry of the cfg-2 UNet frame
(tests/test_gpu_z_properties.py::test_cfg2_stream_batch_rows_are_independent).

Phase 1 (full speed): the frame is run from inputs X and from the row-flipped inputs flip(X) REPS times each; prints how often the
flipped run's output is not the flip of the original's, and how often a run differs from the first run of the same inputs.
Phase 2 (op by op): the plan is replayed one launch at a time for X, X again and flip(X); after every launch every buffer the launch
references (arena buffers, KV caches, static inputs / outputs, GroupNorm accumulators) is hashed per half (= per stream-batch row);
prints the FIRST launches whose hashes are not mirrored (and the launches whose hashes differ between the two X runs).
Phase 3 (full speed, probes inside the plan): an OP_COPY of every launch's output into a side buffer is spliced into the plan
behind it, so that the intermediate results of an UNINTERRUPTED frame can be compared between X and flip(X).

Environment: L2D_WSGEMM / L2D_WSGEMM_NO_TABLE as in the product; REPS; POISON=1 fills every arena buffer and split-K workspace
with NaN bit patterns in front of every frame (stale data then cannot pass for plausible data)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from live2diff_amd import _lib, ops  # noqa: E402
from live2diff_amd.config import sd15_config  # noqa: E402
from live2diff_amd.pipeline_stream_animation_depth import ring_buffer_init, ring_buffer_update  # noqa: E402
from live2diff_amd.unet_hip import HipStreamingUNet  # noqa: E402
from live2diff_amd.weights import device_random_state_dict  # noqa: E402
import bench  # noqa: E402

DEV = torch.device("cuda", 0)
REPS = int(os.environ.get("REPS", "10"))
POISON = os.environ.get("POISON", "0") != "0"
HW = int(os.environ.get("HW", "64"))
PHASES = os.environ.get("PHASES", "123")


def build():
    cfg = sd15_config()
    N, h, w = 2, HW, HW
    unet = HipStreamingUNet(device_random_state_dict(cfg, DEV), cfg, h, w, N)
    g = torch.Generator(device=DEV).manual_seed(1234)
    kv = unet.prepare_cache(N)
    for c in kv:
        c.normal_(generator=g)
    rb = ring_buffer_init(N, cfg.window_size, cfg.sink_size)
    for _ in range(cfg.window_size + 3):
        ring_buffer_update(*rb, cfg.window_size, cfg.sink_size)
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float16)
    i = dict(x=rn(N, 4, 1, h, w), d=rn(N, 4, 1, h, w), enc=rn(N, 77, cfg.cross_attention_dim),
             ts=torch.tensor([399, 199], device=DEV), bias=rb[0].half().to(DEV), pe=rb[1].to(DEV), upd=rb[2].to(DEV))
    return unet, kv, i


def flip_inputs(i):
    return {k: (v.flip(0).contiguous() if v.dim() >= 1 and v.shape[0] == 2 else v) for k, v in i.items()}


def poison(st):
    if not POISON:
        return
    for t in st.arena.all:
        t.view(torch.int16 if t.element_size() == 2 else torch.int32).fill_(-1)      # 0xFFFF / 0xFFFFFFFF: NaN


def step(unet, kv, i, st=None):
    if st is not None:
        poison(st)
    o = unet(i["x"], i["ts"], encoder_hidden_states=i["enc"], temporal_attention_mask=i["bias"], depth_sample=i["d"],
             kv_cache=kv, pe_idx=i["pe"], update_idx=i["upd"])
    torch.cuda.synchronize()
    return o["sample"].clone()


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def load_inputs(unet, st, kv, i):
    """what HipStreamingUNet.__call__ does in front of the plan"""
    N, cfg = unet.N, unet.cfg
    unet._bind_caches(st, kv)
    st.in_sample.copy_(i["x"].reshape(N, cfg.in_channels, -1))
    st.in_depth.copy_(i["d"].reshape(N, cfg.in_channels, -1))
    st.in_t.copy_(i["ts"])
    st.in_enc[:, : st.text_len, : cfg.cross_attention_dim].copy_(i["enc"])
    st.in_bias.copy_(i["bias"]); st.in_pe_idx.copy_(i["pe"]); st.in_upd.copy_(i["upd"])
    st.cond_pl.run()
    st.cond_key = None
    torch.cuda.synchronize()


def halves_hash(t):
    """(hash of first half, hash of second half) of a flat buffer; exact integer arithmetic on the raw bits"""
    v = t.reshape(-1)
    if v.element_size() == 2:
        v = v.view(torch.int16)
    elif v.element_size() == 4:
        v = v.view(torch.int32)
    else:
        v = v.view(torch.int64)
    n = v.numel() // 2
    out = []
    for part in (v[:n], v[n:2 * n]):
        p = part.to(torch.int64)
        wgt = (torch.arange(n, device=p.device, dtype=torch.int64) % 8191) + 1
        out.append((int(p.sum().item()), int((p * wgt).sum().item())))
    return tuple(out)


def main():
    unet, kv, i = build()
    j = flip_inputs(i)
    before = [c.clone() for c in kv]
    before_f = [b.flip(0).contiguous() for b in before]
    print("device", unet.device_name, "| L2D_WSGEMM =", os.environ.get("L2D_WSGEMM", "default"), "| NO_TABLE =",
          os.environ.get("L2D_WSGEMM_NO_TABLE", "0"), "| POISON =", POISON, flush=True)
    a0 = step(unet, kv, i)                 # builds the plan
    st = unet._plans["stream"]
    print("ops per frame:", len(st.pl), flush=True)

    def restore(dst, src):
        for c, b in zip(dst, src):
            c.copy_(b)

    if "1" in PHASES:
        restore(kv, before)
        a0 = step(unet, kv, i, st)
        kv2 = [b.clone() for b in before_f]
        nself_a = nself_b = nflip = 0
        b0 = None
        for rep in range(REPS):
            restore(kv, before)
            a = step(unet, kv, i, st)
            restore(kv2, before_f)
            b = step(unet, kv2, j, st)
            if b0 is None:
                b0 = b
            sa, sb, fl = not torch.equal(a, a0), not torch.equal(b, b0), not torch.equal(b.flip(0), a)
            nself_a += sa; nself_b += sb; nflip += fl
            if sa or sb or fl:
                print(f"  rep {rep}: A != A0: {sa} ({rel(a, a0):.2e})  B != B0: {sb} ({rel(b, b0):.2e})  flip(B) != A: {fl} ({rel(b.flip(0), a):.2e})",
                      flush=True)
        print(f"phase 1: {REPS} reps: A not repeatable {nself_a}, B not repeatable {nself_b}, flip symmetry broken {nflip}", flush=True)
        del kv2

    # ----------------------------------------------------------------------------------------------- phase 2: op by op
    bufs = []          # (name, tensor)
    for k, t in enumerate(st.arena.all):
        bufs.append((f"arena{k}[{t.numel()}x{t.element_size()}]", t))
    for name in ("in_sample", "in_depth", "in_enc", "out_sample", "in_bias"):
        bufs.append((name, getattr(st, name)))
    ranges = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), name, t) for name, t in bufs)

    def lookup(ptr, caches):
        for lo, hi, name, t in ranges:
            if lo <= ptr < hi:
                return name, t
        for k, c in enumerate(caches):
            lo = c.data_ptr()
            if lo <= ptr < lo + c.numel() * 2:
                return f"kv{k}", c
        return None

    arr_t = _lib.L2dOp * 1

    def run_op_by_op(caches, inputs):
        load_inputs(unet, st, caches, inputs)
        poison(st)
        arr = st.pl.array()
        s = _lib.current_stream_ptr()
        rec = []
        for k in range(len(st.pl)):
            one = arr_t.from_address(ctypes.addressof(arr) + k * ctypes.sizeof(_lib.L2dOp))
            _lib.check(_lib.lib.l2d_run_ops(one, 1, ctypes.c_void_p(s)), f"op {k}")
            torch.cuda.synchronize()
            op = st.pl[k]
            hs = {}
            for q in range(16):
                p = op.p[q]
                if not p:
                    continue
                hit = lookup(p, caches)
                if hit is None or hit[0] in hs:
                    continue
                hs[hit[0]] = halves_hash(hit[1])
            g = st.gn_acc
            hs["gn_acc"] = (tuple(g[:, 0].reshape(-1).tolist()), tuple(g[:, 1].reshape(-1).tolist())) if k % 8 == 0 or op.kind == _lib.OP_GN_APPLY else ((), ())
            rec.append(hs)
        return rec, st.out_sample.clone()

    if "2" in PHASES:
        restore(kv, before)
        ra, oa = run_op_by_op(kv, i)
        restore(kv, before)
        ra2, oa2 = run_op_by_op(kv, i)
        kv2 = [b.clone() for b in before_f]
        rb_, ob = run_op_by_op(kv2, j)
        print(f"phase 2: op-by-op: A == A': {torch.equal(oa, oa2)}   flip(B) == A: {torch.equal(ob.flip(0), oa)} ({rel(ob.flip(0), oa):.2e})", flush=True)
        shown = 0
        for k in range(len(st.pl)):
            op = st.pl[k]
            bad_self = [n for n in ra[k] if ra[k][n] != ra2[k].get(n)]
            bad_flip = [n for n in ra[k] if n in rb_[k] and (ra[k][n][0], ra[k][n][1]) != (rb_[k][n][1], rb_[k][n][0])]
            if bad_self or bad_flip:
                print(f"  op {k}: {bench.KIND_NAMES.get(op.kind, op.kind)} {bench.op_dims(op, _lib)} i9..12={list(op.i[9:13])}: "
                      f"not repeatable: {bad_self}  not mirrored: {bad_flip}", flush=True)
                shown += 1
                if shown >= 12:
                    break
        if not shown:
            print("  every launch repeatable and mirrored", flush=True)
        del kv2

    # ----------------------------------------------------------------------------------------------- phase 3: probes in the plan
    if "3" in PHASES:
        # splice a copy of every launch's main output behind it (same stream: the frame stays uninterrupted)
        pl2 = _lib.OpList()
        side = []
        idx2 = []
        for k in range(len(st.pl)):
            op = st.pl[k]
            c = _lib.L2dOp()
            ctypes.memmove(ctypes.byref(c), ctypes.byref(op), ctypes.sizeof(_lib.L2dOp))
            pl2.append(c)
            idx2.append(len(pl2) - 1)
            q = {_lib.OP_IGEMM: 6, _lib.OP_WSGEMM: 6, _lib.OP_ROWGEMM: 4, _lib.OP_PCONV: 6, _lib.OP_GN_APPLY: 5, _lib.OP_LAYERNORM: 3,
                 _lib.OP_FLASH_ATTN: 3, _lib.OP_TATTN_STREAM: 8}.get(op.kind)
            if q is None or not op.p[q]:
                continue
            hit = lookup(op.p[q], kv)
            if hit is None or hit[0].startswith("kv"):
                continue
            t = hit[1]
            sd = torch.empty_like(t)
            side.append((k, hit[0], sd))
            cp, keep = ops.copy(t, sd, t.numel() * t.element_size())
            pl2.append(cp, *keep)
        print(f"phase 3: {len(pl2)} launches ({len(side)} probes)", flush=True)

        def run_probed(caches, inputs):
            load_inputs(unet, st, caches, inputs)
            for tag, _ in st.tattn_ops:          # the cache pointers _bind_caches has just bound in st.pl
                pl2[idx2[tag]].p[1] = st.pl[tag].p[1]
            pl2._arr = None
            poison(st)
            pl2.run()
            torch.cuda.synchronize()
            return [sd.clone() for _, _, sd in side], st.out_sample.clone()

        nbad = 0
        for rep in range(REPS):
            restore(kv, before)
            sa, oa = run_probed(kv, i)
            kv2 = [b.clone() for b in before_f]
            sb, ob = run_probed(kv2, j)
            del kv2
            ok = torch.equal(ob.flip(0), oa)
            if ok:
                continue
            nbad += 1
            shown = 0
            for (k, name, _), ta, tb in zip(side, sa, sb):
                n = ta.numel() // 2
                va, vb = ta.reshape(-1), tb.reshape(-1)
                if torch.equal(va[:n], vb[n:2 * n]) and torch.equal(va[n:2 * n], vb[:n]):
                    continue
                op = st.pl[k]
                d = (torch.cat([vb[n:2 * n], vb[:n]]).float() - va[:2 * n].float())
                nz = (d != 0).nonzero().reshape(-1)
                ld = op.i[15] if op.kind in (_lib.OP_IGEMM, _lib.OP_WSGEMM) and op.i[15] > 0 else 1
                rows = sorted(set((nz // ld).tolist()))
                cols = sorted(set((nz % ld).tolist()))
                print(f"  rep {rep}: first asymmetric launch: op {k} {bench.KIND_NAMES.get(op.kind, op.kind)} {bench.op_dims(op, _lib)} "
                      f"i9..12={list(op.i[9:13])} i19..21={list(op.i[19:22])}: {nz.numel()} elements, max |d| {d.abs().max().item():.4g}, "
                      f"rows {rows[:12]}{'...' if len(rows) > 12 else ''} ({len(rows)}) cols {cols[:12]}{'...' if len(cols) > 12 else ''} ({len(cols)})",
                      flush=True)
                shown += 1
                if shown >= 3:
                    break
        print(f"phase 3: flip symmetry broken in {nbad} / {REPS} probed frames", flush=True)


if __name__ == "__main__":
    main()
