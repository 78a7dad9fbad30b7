"""Multi-GPU rendering on one node: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The scene is read-only and replicated on every GPU; only the framebuffer is exchanged,
once per frame, by one all-reduce:

  tile sharding    rank r renders the 32x32 pixel tiles t with t % world == r into a zeroed
                   buffer; SUM over ranks reassembles the image exactly (x + 0 = x), so the
                   result is bit-identical to a one-GPU render.
  sample sharding  rank r renders samples [r*spp, (r+1)*spp) of every pixel as raw sums;
                   SUM over ranks then a division by the summed sample count.  Summation
                   order differs from a one-GPU render, so equality is to fp32 rounding.

The reference has no multi-GPU path (one Unity process, one GPU); this is the exchange
step its per-frame dispatch (Assets/Script/RenderPass/RayTracingPass.cs:101) would need
when the image is partitioned.
"""
import torch
import torch.distributed as dist

from . import types as T


def tile_shard_kwargs(rank, world):
    """Extension-parameter overrides that make a render cover only this rank's tiles."""
    return dict(shard_rank=rank, shard_count=world) if world > 1 else {}


def sample_shard_kwargs(rank, world, spp_per_rank, flags=0):
    """Extension-parameter overrides for this rank's sample range (raw sums out)."""
    kw = dict(sample_offset=rank * spp_per_rank, spp=spp_per_rank)
    kw["flags"] = flags | (T.FLAG_OUTPUT_SUM if world > 1 else 0)
    return kw


def combine_tiles(buffers, group=None):
    """All-reduce (SUM) tensors whose un-owned pixels are zero.  In place; returns them."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for b in buffers:
            if b is not None:
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
    return buffers


def combine_samples(sum_rgba, group=None):
    """All-reduce (SUM) a (h,w,4) tensor of rgb sums + sample count in alpha, then normalise
    in place to (mean rgb, 1)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sum_rgba, op=dist.ReduceOp.SUM, group=group)
    n = sum_rgba[..., 3:4].clone()
    sum_rgba.div_(torch.where(n > 0, n, torch.ones_like(n)))
    return sum_rgba


def render_tiles(execute, frame, width, height, rank, world, device, want_ids=True, group=None, **ext_kw):
    """Tile-sharded render.  `execute(frame, w, h, out_rgba=, out_prim_id=, **ext)` is
    RayTracingPass.Execute (or any callable with that signature).  Returns (rgba, ids)
    torch tensors holding the full image on every rank."""
    rgba = torch.zeros((height, width, 4), dtype=torch.float32, device=device)
    ids = torch.zeros((height, width, 2), dtype=torch.int32, device=device) if want_ids else None
    execute(frame, width, height, out_rgba=rgba, out_prim_id=ids, **tile_shard_kwargs(rank, world), **ext_kw)
    combine_tiles([rgba, ids], group)
    return rgba, ids


def render_samples(execute, frame, width, height, rank, world, device, spp_per_rank, group=None, **ext_kw):
    """Sample-sharded path-traced render.  Returns the normalised image on every rank."""
    rgba = torch.zeros((height, width, 4), dtype=torch.float32, device=device)
    kw = dict(ext_kw)
    kw.update(sample_shard_kwargs(rank, world, spp_per_rank, ext_kw.get("flags", 0)))
    kw["mode"] = T.MODE_PT
    execute(frame, width, height, out_rgba=rgba, **kw)
    if world > 1:
        combine_samples(rgba, group)
    return rgba
