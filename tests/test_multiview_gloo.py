"""world_size-2 gloo test (CPU) of the per-camera sharding + all-gather host logic."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(cam):
    H, W = 6, 8
    base = torch.arange(5 * H * W, dtype=torch.float32).view(5, H, W) + 1000.0 * cam
    return base[:3], base[3:4], base[4:5]


def _worker(rank, world, port, n_views, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "guidedvd-3dgs_amd"))
    import multiview
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = multiview.render_views_sharded(_fake_render, list(range(n_views)))
    ok = out.shape[0] == n_views
    for v in range(n_views):
        c, d, a = _fake_render(v)
        ok = ok and torch.equal(out[v], torch.cat([c, d, a], 0))
    q.put((rank, bool(ok), multiview.shard_views(n_views, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_views, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_sharded_render_allgather_even():
    res = _run(6)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == [0, 2, 4] and res[1][2] == [1, 3, 5]


def test_sharded_render_allgather_uneven_25_views():
    res = _run(25)  # the guidance-render count of one diffusion run; 13/12 split
    assert all(ok for _, ok, _ in res)
    assert len(res[0][2]) == 13 and len(res[1][2]) == 12


def test_single_view_fewer_than_ranks():
    res = _run(1)
    assert all(ok for _, ok, _ in res)


def _grad_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "guidedvd-3dgs_amd"))
    import multiview
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    P = 37
    base = [torch.randn(P, 3, generator=g), torch.randn(P, 16, 3, generator=g), torch.randn(P, 1, generator=g)]
    views = [torch.randn(P, 3, generator=g) for _ in range(world)]

    def loss(params, v):   # stand-in for render(view v) + photometric loss
        out = (params[0] * views[v]).sum() ** 2 + (params[1].sum(-1).sum(-1) * views[v][:, 0]).sum()
        return out + params[2].sum() if v == 0 else out

    full = [b.clone().requires_grad_(True) for b in base]
    sum(loss(full, v) for v in range(world)).backward()
    mine = [b.clone().requires_grad_(True) for b in base]
    frozen = torch.zeros(3, requires_grad=False)
    loss(mine, rank).backward()
    if rank != 0:
        assert mine[2].grad is None            # this rank's view does not touch the third parameter
    nbytes = multiview.allreduce_gradients(mine + [frozen])
    ok = nbytes == 4 * sum(b.numel() for b in base)
    for a, b in zip(mine, full):
        ok = ok and torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)
    q.put((rank, bool(ok), None))
    dist.barrier()
    dist.destroy_process_group()


def test_training_step_gradient_allreduce_over_two_views():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
