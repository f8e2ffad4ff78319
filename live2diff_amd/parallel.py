"""Multi-GPU support: one process per GPU, every rank an INDEPENDENT frame stream.

The reference has no distributed path at all.  Streams share nothing per frame --
weights are read-only replicas, KV-cache / latent buffers / ring indices are private -- so the only
collectives are (a) a one-time RCCL replication of the PACKED weights from rank 0 (which runs the packing pass once;
replicate_packed_weights), done as scatter (1/G shard per peer, one xGMI link each) + all-gather (all links), in a few flat
1 GB buckets, with an all-reduced checksum (broadcast_state_dict does the same for a raw fp16 state dict),
and (b) the barrier / max / all-gather of per-rank results around the timed region of the benchmark.  Nothing on
the per-frame path.

`backend` is "nccl" (= RCCL on ROCm) on GPUs and "gloo" in the CPU tests.
"""
import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, force: bool = False):
    """Initialise from torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*). Returns (rank, world, local).
    `force`: create the process group even for world size 1 (exercises the backend's API surface on a single GPU)."""
    # RCCL between processes shares device buffers through IPC handles; the host driver of these nodes supports dmabuf IPC only
    # (without this the first collective fails with `hipIpcGetMemHandle: invalid argument`).  Must be in the environment before the
    # HIP runtime of this process creates its first context, i.e. before the first torch.cuda call below -- ranks started by
    # `torchrun` directly get it here, not only through bench.py's self-launcher.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_state_dict(spec: Dict[str, tuple], sd: Optional[Dict[str, torch.Tensor]], device, dtype=torch.float16,
                         src: int = 0, bucket_elems: int = 512 * 1024 * 1024, algo: str = "scatter_allgather",
                         force_collectives: bool = False) -> Dict[str, torch.Tensor]:
    """Rank `src` holds `sd`; every rank returns a full copy on `device`.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a root-sourced broadcast is bound by what ONE link / a
    ring step carries.  Default algorithm, per flat bucket of up to `bucket_elems` elements (1 GB fp16: the 2.56 GB UNet
    is 3 buckets instead of 1 222 tensors):
        1. scatter     rank `src` sends shard g (1/G of the bucket) to peer g -- G-1 different shards leave the root over
                       G-1 different links at once, each link carries 1/G of the bytes;
        2. all-gather  every rank re-sends its shard to all peers (all links of all GPUs busy).
    `algo="broadcast"` keeps the plain bucketed `dist.broadcast` for comparison.  A position-sensitive checksum (fp64,
    every element weighted by a function of its index in the bucket, so shards gathered in the wrong rank order change it)
    is all-reduced with MIN and MAX, which must agree, to verify the replicas.  `force_collectives` runs the collectives
    even for world size 1 (a one-rank process group: the RCCL calls execute on a single-GPU box)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1 and not (force_collectives and dist.is_initialized()):
        return {k: sd[k].to(device=device, dtype=dtype) for k in spec}
    keys = list(spec)
    out: Dict[str, torch.Tensor] = {}
    i = 0
    checksum = torch.zeros(1, dtype=torch.float64, device=device)
    while i < len(keys):
        j, n = i, 0
        while j < len(keys) and (n == 0 or n + _numel(spec[keys[j]]) <= bucket_elems):
            n += _numel(spec[keys[j]])
            j += 1
        shard = (n + world - 1) // world
        shard = (shard + 7) // 8 * 8                       # 16-byte aligned shards
        flat = torch.empty(shard * world, dtype=dtype, device=device)
        if rank == src:
            off = 0
            for k in keys[i:j]:
                m = _numel(spec[k])
                flat[off:off + m].copy_(sd[k].reshape(-1).to(device=device, dtype=dtype))
                off += m
            flat[n:].zero_()
        if algo == "broadcast":
            dist.broadcast(flat, src=src)
        else:
            mine = torch.empty(shard, dtype=dtype, device=device)
            dist.scatter(mine, scatter_list=(list(flat.view(world, shard).unbind(0)) if rank == src else None), src=src)
            _all_gather_flat(flat, mine, world, shard)
        checksum += _checksum(flat, n)
        off = 0
        for k in keys[i:j]:
            m = _numel(spec[k])
            out[k] = flat[off:off + m].view(spec[k])
            off += m
        i = j
    lo, hi = checksum.clone(), checksum.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"weight broadcast checksum mismatch across ranks: {lo.item()} vs {hi.item()}")
    return out


def replicate_tensors(tensors: Optional[Dict[str, torch.Tensor]], device, src: int = 0, bucket_bytes: int = 1 << 30,
                      force_collectives: bool = False) -> Dict[str, torch.Tensor]:
    """Rank `src` holds `tensors` (any dtypes: the PACKED weights are fp16 matrices, fp32 biases / column sums); every rank returns
    a full copy on `device`.  Names, shapes and dtypes travel as one small object broadcast; the payload as raw bytes, every tensor
    at a 16-byte aligned offset of a flat uint8 bucket of up to `bucket_bytes`, each bucket through the same scatter (1/G shard per
    peer, one xGMI link each) + all-gather (all links) as broadcast_state_dict, verified by the same position-weighted checksum
    (over the bytes), all-reduced MIN / MAX."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1 and not (force_collectives and dist.is_initialized()):
        return {k: v.to(device) for k, v in tensors.items()}
    box = [[(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in tensors.items()] if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    items = box[0]
    esize = {dt: torch.empty(0, dtype=getattr(torch, dt)).element_size() for dt in {it[2] for it in items}}
    size = lambda shp, dt: (_numel(shp) * esize[dt] + 15) // 16 * 16          # bytes of a tensor's slot in a bucket (16-byte aligned)
    out: Dict[str, torch.Tensor] = {}
    checksum = torch.zeros(1, dtype=torch.float64, device=device)
    # (zero-byte tensors never enter a bucket: they are emitted directly; a bucket is never empty)
    for k, shp, dt in items:
        if _numel(shp) == 0:
            out[k] = torch.empty(shp, dtype=getattr(torch, dt), device=device)
    items = [it for it in items if _numel(it[1]) > 0]
    i = 0
    while i < len(items):
        j, n = i, 0
        while j < len(items) and (j == i or n + size(items[j][1], items[j][2]) <= bucket_bytes):
            n += size(items[j][1], items[j][2])
            j += 1
        shard = ((n + world - 1) // world + 15) // 16 * 16
        flat = torch.empty(shard * world, dtype=torch.uint8, device=device)
        if rank == src:
            flat.zero_()
            off = 0
            for k, shp, dt in items[i:j]:
                nb = _numel(shp) * esize[dt]               # straight into the flat view: no intermediate device copy beside the bucket
                flat[off:off + nb].view(getattr(torch, dt)).view(shp).copy_(tensors[k])
                off += size(shp, dt)
        mine = torch.empty(shard, dtype=torch.uint8, device=device)
        dist.scatter(mine, scatter_list=(list(flat.view(world, shard).unbind(0)) if rank == src else None), src=src)
        _all_gather_flat(flat, mine, world, shard)
        checksum += _checksum(flat, n)
        off = 0
        for k, shp, dt in items[i:j]:
            nb = _numel(shp) * esize[dt]
            out[k] = flat[off:off + nb].view(getattr(torch, dt)).view(shp)
            off += size(shp, dt)
        i = j
    lo, hi = checksum.clone(), checksum.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"packed-weight replication checksum mismatch across ranks: {lo.item()} vs {hi.item()}")
    return out


def replicate_packed_weights(unet_or_none, device, src: int = 0, force_collectives: bool = False):
    """"one-time RCCL broadcast of PACKED fp16 weights".  Rank `src` passes its HipStreamingUNet (it packed the state
    dict once); every rank gets a `PackedWeights` to construct its own instance from (`HipStreamingUNet(packed, cfg, h, w, N)`) --
    no rank but `src` runs the packing pass.  Rank `src` gets its own tensors back untouched."""
    from .unet_hip import PackedWeights
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1 and not (force_collectives and dist.is_initialized()):
        return unet_or_none.packed_state()
    box = [unet_or_none._packed_meta() if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    W = replicate_tensors(unet_or_none.W if rank == src else None, device, src=src, force_collectives=force_collectives)
    if rank == src:
        return unet_or_none.packed_state()
    return PackedWeights(W, box[0])


_CK_CHUNK = 1 << 24


def _checksum(flat: torch.Tensor, n: int) -> torch.Tensor:
    """sum_i flat[i] * (1 + (i mod 8191) / 8192) in fp64, in 16 M-element slices (no bucket-sized fp64 temporary).  The
    weight depends on the element's position, so two equal-sized shards that swapped places change the sum."""
    total = torch.zeros(1, dtype=torch.float64, device=flat.device)
    for a in range(0, n, _CK_CHUNK):
        b = min(n, a + _CK_CHUNK)
        wgt = torch.arange(a, b, device=flat.device, dtype=torch.int64).remainder_(8191).to(torch.float64).mul_(1.0 / 8192.0).add_(1.0)
        total += torch.sum(flat[a:b].to(torch.float64) * wgt, dtype=torch.float64)
    return total


def _all_gather_flat(flat: torch.Tensor, mine: torch.Tensor, world: int, shard: int):
    try:
        dist.all_gather_into_tensor(flat, mine)
    except (RuntimeError, NotImplementedError):            # a backend without the flat form (older gloo)
        dist.all_gather(list(flat.view(world, shard).unbind(0)), mine)


def gather_floats(x: float, device="cpu"):
    """Every rank's value, in rank order (per-rank frames/s for the aggregate report); [x] without a process group."""
    if not dist.is_initialized():
        return [x]
    world = dist.get_world_size()
    t = torch.tensor([x], dtype=torch.float64, device=device)
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device="cpu") -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device="cpu") -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
