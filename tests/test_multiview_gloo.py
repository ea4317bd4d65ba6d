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


def _gpu_render_worker(rank, world, port, q):
    try:
        import sys
        import numpy as np
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (root, os.path.join(root, "guidedvd-3dgs_amd"), os.path.join(root, "tests")):
            sys.path.insert(0, p)
        import multiview
        import synthetic as syn
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sc = syn.scene_c2(P=3000, W=160, H=120, n_cams=5)
        t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, requires_grad=rg)
        P = sc["means3D"].shape[0]
        prm = [t(sc[k], True) for k in ("means3D", "opacities", "scales", "rotations", "shs")]
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)

        def settings(cam):
            return GaussianRasterizationSettings(image_height=120, image_width=160, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
                                                 bg=t(sc["bg"]), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]),
                                                 projmatrix=t(cam["projmatrix"]), sh_degree=3, campos=t(cam["campos"]),
                                                 prefiltered=False, debug=False, confidence=torch.ones((P, 1), device=dev))

        def render(cam):
            c, _, d, a = GaussianRasterizer(settings(cam))(means3D=prm[0], means2D=m2, opacities=prm[1], shs=prm[4], scales=prm[2],
                                                           rotations=prm[3])
            return c, d, a

        with torch.no_grad():
            gathered = multiview.render_views_sharded(render, sc["cameras"])
            ok = gathered.shape == (5, 5, 120, 160)
            for v, cam in enumerate(sc["cameras"]):
                c, d, a = render(cam)
                ok = ok and torch.equal(gathered[v], torch.cat([c, d, a], 0))
        # two-view training step: each rank differentiates ITS view, gradients are summed with one all-reduce
        gC = torch.randn(3, 120, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        for p_ in prm + [m2]:
            p_.grad = None
        (render(sc["cameras"][rank])[0] * gC).sum().backward()
        multiview.allreduce_gradients(prm + [m2])
        mine = [p_.grad.clone() for p_ in prm]
        for p_ in prm + [m2]:
            p_.grad = None
        sum((render(sc["cameras"][v])[0] * gC).sum() for v in range(world)).backward()
        for a_, p_ in zip(mine, prm):
            ok = ok and torch.allclose(a_, p_.grad, rtol=1e-5, atol=1e-7 * float(p_.grad.abs().max()))
        q.put((rank, bool(ok), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, False, traceback.format_exc()))
        raise


import pytest  # noqa: E402


@pytest.mark.gpu
def test_view_shards_and_two_view_gradient_allreduce_on_device_two_ranks_one_gpu():
    """The per-camera render sharding + all-gather and the two-view training-step gradient all-reduce with the real
    rasterizer, two ranks sharing the one GPU (gloo carries the collectives)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_render_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, ok, tb in res:
        assert ok, (rank, tb)
