"""Multi-GPU support: one process per GPU, every rank an INDEPENDENT frame stream.

The reference has no distributed path at all.  Streams share nothing per frame --
weights are read-only replicas, KV-cache / latent buffers / ring indices are private -- so the only
collectives are (a) a one-time RCCL broadcast of the fp16 weights from rank 0 (flattened into a few large
messages: xGMI is per-link bound, big transfers amortise the ring latency), with an all-reduced checksum,
and (b) the barrier / max / sum around the timed region of the benchmark.  Nothing on the per-frame path.

`backend` is "nccl" (= RCCL on ROCm) on GPUs and "gloo" in the CPU tests.
"""
import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None):
    """Initialise from torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_state_dict(spec: Dict[str, tuple], sd: Optional[Dict[str, torch.Tensor]], device, dtype=torch.float16,
                         src: int = 0, bucket_elems: int = 256 * 1024 * 1024) -> Dict[str, torch.Tensor]:
    """Rank `src` holds `sd`; every rank returns a full copy on `device`.  Tensors are packed into flat buckets of
    up to `bucket_elems` elements (512 MB fp16) -> ~5 broadcasts for the 2.56 GB UNet instead of 1 222 small ones.
    A checksum (sum of per-bucket fp64 sums) is all-reduced (MIN == MAX) to verify the replicas."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
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
        flat = torch.empty(n, dtype=dtype, device=device)
        if rank == src:
            off = 0
            for k in keys[i:j]:
                m = _numel(spec[k])
                flat[off:off + m].copy_(sd[k].reshape(-1).to(device=device, dtype=dtype))
                off += m
        dist.broadcast(flat, src=src)
        checksum += flat.double().sum()
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
