"""Per-camera sharding of inference renders over ranks.

The guidance renders of a diffusion run (train_guidedvd.py:521-527: 25 views), the trajectory-pool
renders (:159-164) and eval renders are independent units: every rank holds a replica of the
Gaussians, renders the views `rank, rank+world, ...` and ONE all-gather (RCCL on GPUs; gloo in the
CPU tests) returns all images [n_views, 5, H, W] = (rgb, depth, alpha) on every rank.  xGMI is
point-to-point, so a single fused all-gather of the 5-channel stack (154 MB for 25x640x480) is
preferred over per-image or per-channel collectives.
"""
import torch
import torch.distributed as dist


def shard_views(n_views, rank, world):
    """Round-robin view ids owned by `rank`."""
    return list(range(rank, n_views, world))


def gather_views(local, n_views, group=None):
    """local: [k, C, H, W] holding this rank's views in shard_views order (k may differ by one
    between ranks).  Returns [n_views, C, H, W] in view order on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local[:n_views]
    rank = dist.get_rank(group)
    kmax = (n_views + world - 1) // world
    pad = local.new_zeros((kmax,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    out = local.new_empty((world * kmax,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    out = out.view(world, kmax, *local.shape[1:])
    # rank r, slot s holds view s*world + r
    res = out.permute(1, 0, *range(2, out.dim())).reshape(world * kmax, *local.shape[1:])
    assert shard_views(n_views, rank, world) == list(range(rank, n_views, world))
    return res[:n_views].contiguous()


def render_views_sharded(render_fn, cameras, group=None):
    """render_fn(camera) -> (color[3,H,W], depth[1,H,W], alpha[1,H,W]).  Returns [n,5,H,W] on all ranks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_views(len(cameras), rank, world)
    imgs = []
    for v in mine:
        c, d, a = render_fn(cameras[v])
        imgs.append(torch.cat([c, d, a], 0))
    if imgs:
        local = torch.stack(imgs, 0)
    else:
        c, d, a = render_fn(cameras[0])  # shape probe for ranks that own no view
        local = torch.cat([c, d, a], 0).new_zeros((0, 5) + tuple(c.shape[1:]))
    return gather_views(local, len(cameras), group)


def allreduce_gradients(params, group=None):
    """SURVEY 8e row 2 (training step, width 2: train view on one GPU, pseudo view on the other, train_guidedvd.py:334,357):
    sum the per-Gaussian gradients of `params` over the ranks with ONE all-reduce of a flat fp32 bucket
    (~62 floats per Gaussian: 49.6 MB at 200k) instead of one collective per tensor -- xGMI all-reduce at this size is
    latency-dominated.  A parameter without a gradient on this rank contributes zeros.  In place; returns the bucket
    size in bytes.  Densify / prune decisions taken from the summed gradients are then identical on every replica."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    dev = params[0].device
    flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    off = 0
    for p in params:
        if p.grad is not None:
            flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        off += p.numel()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()
    return flat.numel() * 4
