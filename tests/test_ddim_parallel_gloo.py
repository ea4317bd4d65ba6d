"""Multi-process (gloo, CPU) tests of the DDIM step's multi-GPU decomposition (lvdm_amd/parallel.py):
CFG pair x frame shards must reproduce the single-process step -- plain and guided (distributed x-gradient) -- for
even and uneven splits.  The CPU path runs the explicit reference math (ops.use_reference_math), so what is under
test is the partitioning / exchange logic: all-to-all re-sharding around the temporal layers, shard-group GroupNorm
statistics, e_t all-gather, x-gradient all-reduce."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SMALL_UNET = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[1, 2], num_res_blocks=1,
                  channel_mult=[1, 2], dropout=0.0, num_head_channels=64, transformer_depth=1, context_dim=64,
                  use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True,
                  temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False, temporal_length=16,
                  addition_attention=True, image_cross_attention=True, default_fs=10, fs_condition=True)
SMALL_VAE = dict(double_z=True, z_channels=4, resolution=16, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2],
                 num_res_blocks=1, attn_resolutions=[], dropout=0.0)
T, HL, WL = 5, 8, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fill_by_name import fill_by_name
    from lvdm_amd import ops
    from lvdm_amd.model import LatentDiffusion
    ops.use_reference_math(True)
    torch.manual_seed(0)
    ld = LatentDiffusion(SMALL_UNET, SMALL_VAE)
    fill_by_name(ld, std=0.08)
    ld = ld.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, T, HL, WL, generator=g)
    cond = {"c_crossattn": [torch.randn(1, 77 + 16, 64, generator=g)], "c_concat": [torch.randn(1, 4, T, HL, WL, generator=g) * 0.2]}
    uc = {"c_crossattn": [torch.randn(1, 77 + 16, 64, generator=g)], "c_concat": cond["c_concat"]}
    noise = torch.randn(1, 4, T, HL, WL, generator=g)
    renoise = torch.randn(1, 4, T, HL, WL, generator=g)
    gimgs = torch.rand(T, 3, HL * 2, WL * 2, generator=g)
    gmask = (torch.rand(T, 1, HL * 2, WL * 2, generator=g) > 0.3).float()
    return ld, x, cond, uc, noise, renoise, gimgs, gmask


def _steps(ld, x, cond, uc, noise, renoise, gimgs, gmask, plan):
    from lvdm_amd.guidance import LossGuidance
    from lvdm_amd.samplers import DDIMSampler, DDIMSamplerGuidance
    fs = torch.tensor([10])
    out = {}
    s = DDIMSampler(ld)
    s.parallel = plan
    s.make_schedule(50, "uniform_trailing", 1.0)
    index = 30
    t = torch.full((1,), int(s.ddim_timesteps[index]), dtype=torch.long)
    with torch.no_grad():
        out["plain_xprev"], out["plain_x0"] = s.p_sample_ddim(x, cond, t, index=index, unconditional_guidance_scale=7.5,
                                                              unconditional_conditioning=uc, guidance_rescale=0.7, fs=fs,
                                                              noise=noise)
    sg = DDIMSamplerGuidance(ld)
    sg.parallel = plan
    sg.make_schedule(50, "uniform_trailing", 1.0)
    lg = LossGuidance(ddim_steps=50, recur_steps=1, device="cpu")
    lg.set_hw(HL * 2, WL * 2)
    lg.set_guidance_images(gimgs)
    lg.set_guidance_masks(gmask)
    out["guided_xprev"], out["guided_x0"] = sg.p_sample_ddim(x, cond, t, index=index, unconditional_guidance_scale=7.5,
                                                             unconditional_conditioning=uc, guidance_rescale=0.7, fs=fs,
                                                             loss_guidance_fn=lg, noise=noise, renoise=renoise)
    return out


def _worker(rank, world, cfg, port, q):
    try:
        import torch.distributed as dist
        torch.set_num_threads(2)
        built = _build()
        from lvdm_amd import parallel
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ref = _steps(*built, plan=None)                       # every rank: the single-process step
        plan = parallel.ParallelPlan(T, cfg=cfg)
        got = _steps(*built, plan=plan)
        errs = {}
        for k in ref:
            scale = float(ref[k].abs().max())
            errs[k] = float((got[k] - ref[k]).abs().max()) / scale
        # replicated draws: with rank-dependent global RNG state (the common seed + rank convention) the x_T / sigma noise /
        # re-noise of the samplers must still be identical on every rank -- they come from the plan's broadcast-seeded generator
        torch.manual_seed(1000 + rank)
        from lvdm_amd.samplers import DDIMSampler
        sr = DDIMSampler(built[0])
        sr.parallel = plan
        for _ in range(3):
            plan.check_replicated(sr._randn((1, 4, T, HL, WL), torch.device("cpu")), "sampler noise under a plan")
        # user seeding governs (advisor finding, round 2): every plan's stream is a function of rank 0's torch.manual_seed (here
        # the 0 of _build()), whatever the other ranks were seeded with -- but never the global generator's own stream (advisor
        # finding, round 3: seeded with initial_seed() itself, x_T would repeat an earlier global draw of the same size, e.g. the
        # VAE posterior sample of encode_first_stage)
        assert plan.seed == parallel.plan_seed(0, 0) and plan.seed not in (0, 1000 + rank), plan.seed
        first = torch.randn((1, 4, T, HL, WL), generator=torch.Generator().manual_seed(plan.seed))
        torch.manual_seed(0)
        assert not torch.equal(first, torch.randn((1, 4, T, HL, WL)))
        torch.manual_seed(777 + 13 * rank)
        plan2 = parallel.ParallelPlan(T, cfg=cfg)
        assert plan2.seed == parallel.plan_seed(777, 1) != parallel.plan_seed(777, 0), plan2.seed
        plan2.check_replicated(torch.randn((5, 7), generator=plan2.generator("cpu")), "second plan's noise")
        # the guidance term must actually be exercised: guided != plain
        moved = float((ref["guided_xprev"] - ref["plain_xprev"]).abs().max())
        q.put((rank, errs, moved, (plan.cfg, plan.F, plan.cfg_rank, plan.frame_rank, plan.shard.counts)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent instead of a silent non-zero exit
        import traceback
        q.put((rank, {"exception": traceback.format_exc()}, 0.0, None))
        raise e


def _run(world, cfg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, cfg, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert "exception" not in r[1], r[1]["exception"]
    for p in procs:
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("world,cfg", [(2, 2), (2, 1), (3, 1), (4, 2)])
def test_cfg_pair_and_frame_shards_reproduce_the_single_process_step(world, cfg):
    """fp32 on CPU: the partitioned step differs from the single-process one only by summation order
    (GroupNorm partial sums, gradient accumulation over branches) -> 2e-5 relative to the tensor's max."""
    res = _run(world, cfg)
    layouts = [r[3] for r in res]
    assert [(l[2], l[3]) for l in layouts] == [(r // (world // cfg), r % (world // cfg)) for r in range(world)]
    assert all(sum(l[4]) == T and max(l[4]) - min(l[4]) <= 1 for l in layouts)
    for rank, errs, moved, _ in res:
        assert moved > 1e-3, "guidance gradient did not change the step: test is vacuous"
        for k, e in errs.items():
            assert e < 2e-5, (rank, k, e)


def test_split_counts_and_resharding_geometry():
    sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
    from lvdm_amd import parallel
    assert parallel.split_counts(25, 4) == [7, 6, 6, 6]
    assert parallel.split_counts(25, 1) == [25]
    assert parallel.split_counts(9216, 4) == [2304] * 4
    assert sum(parallel.split_counts(2240, 3)) == 2240


def test_batched_cfg_evaluation_equals_the_two_sequential_calls():
    """DDIMSampler.batch_cfg: one batch-2 U-Net call vs the reference's two batch-1 calls (ddim.py:222-223), fp32 CPU."""
    built = _build()   # switches the explicit-math CPU path on for this process: restored below
    ld, x, cond, uc, noise = built[:5]
    from lvdm_amd import ops
    from lvdm_amd.samplers import DDIMSampler
    outs = []
    for flag in (False, True):
        s = DDIMSampler(ld)
        s.batch_cfg = flag
        s.make_schedule(50, "uniform_trailing", 1.0)
        t = torch.full((1,), int(s.ddim_timesteps[30]), dtype=torch.long)
        with torch.no_grad():
            outs.append(s.p_sample_ddim(x, cond, t, index=30, unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                        guidance_rescale=0.7, fs=torch.tensor([10]), noise=noise))
    ops.use_reference_math(False)
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).abs().max()) < 2e-5 * float(a.abs().max())


# ---------------------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, q, cfg=2):
    """Two processes sharing cuda:0, collectives over gloo (device tensors staged through the host): exercises the
    CFG-pair decomposition with the real HIP kernels (f16 U-Net, flash attention fwd/bwd, GroupNorm/LayerNorm/GEGLU
    kernels under autograd) on the one GPU the test box has."""
    try:
        import torch.distributed as dist
        sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from fill_by_name import fill_by_name
        from lvdm_amd import parallel
        from lvdm_amd.guidance import LossGuidance
        from lvdm_amd.model import LatentDiffusion
        from lvdm_amd.samplers import DDIMSampler, DDIMSamplerGuidance
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.manual_seed(0)
        ld = fill_by_name(LatentDiffusion(SMALL_UNET, SMALL_VAE), std=0.08).eval().to(dev)
        ld.model.diffusion_model.half().to_token_major()
        # token-major VAE, as the product path runs it: its latent gradient comes back channels-last, so the collectives meet the
        # rank-dependent strides they meet in production (an un-padded frame shard keeps them through FrameShard.gather, a padded
        # one does not)
        ld.first_stage_model.half().to_token_major()
        ld.requires_grad_(False)
        orig_apply = ld.apply_model
        ld.apply_model = lambda x, t, c, **kw: orig_apply(x.half(), t, {k: [v.half() for v in vs] for k, vs in c.items()}, **kw)
        orig_dec = ld.decode_core
        ld.decode_core = lambda z, **kw: orig_dec(z.half(), **kw)
        g = torch.Generator().manual_seed(5)
        mk = lambda *s: torch.randn(*s, generator=g).to(dev)
        x = mk(1, 4, T, HL, WL)
        cond = {"c_crossattn": [mk(1, 93, 64)], "c_concat": [mk(1, 4, T, HL, WL) * 0.2]}
        uc = {"c_crossattn": [mk(1, 93, 64)], "c_concat": cond["c_concat"]}
        noise, renoise = mk(1, 4, T, HL, WL), mk(1, 4, T, HL, WL)
        gimgs = torch.rand(T, 3, HL * 2, WL * 2, generator=g).to(dev)
        fs = torch.tensor([10], device=dev)
        outs = {}
        for tag, plan in (("single", None), ("pair", parallel.ParallelPlan(T, cfg=cfg))):
            s = DDIMSampler(ld)
            s.parallel = plan
            s.make_schedule(50, "uniform_trailing", 1.0)
            t = torch.full((1,), int(s.ddim_timesteps[30]), dtype=torch.long, device=dev)
            with torch.no_grad():
                p_x, _ = s.p_sample_ddim(x, cond, t, index=30, unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                         guidance_rescale=0.7, fs=fs, noise=noise)
            sg = DDIMSamplerGuidance(ld)
            sg.parallel = plan
            sg.make_schedule(50, "uniform_trailing", 1.0)
            lg = LossGuidance(ddim_steps=50, recur_steps=1, device=str(dev))
            lg.set_hw(HL * 2, WL * 2)
            lg.set_guidance_images(gimgs)
            g_x, _ = sg.p_sample_ddim(x, cond, t, index=30, unconditional_guidance_scale=7.5, unconditional_conditioning=uc,
                                      guidance_rescale=0.7, fs=fs, loss_guidance_fn=lg, noise=noise, renoise=renoise)
            outs[tag] = (p_x.float().cpu(), g_x.float().cpu())
        errs = [float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(outs["pair"], outs["single"])]
        moved = float((outs["single"][1] - outs["single"][0]).abs().max())
        q.put((rank, {"plain": errs[0], "guided": errs[1]}, moved, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {"exception": traceback.format_exc()}, 0.0, None))
        raise e


@pytest.mark.gpu
@pytest.mark.parametrize("world,cfg", [(2, 2), (2, 1), (4, 2)])
def test_ranks_on_one_gpu_with_hip_kernels(world, cfg):
    """(2, 2): CFG pair; (2, 1): two frame shards (5 frames -> 3 + 2: all-to-all re-sharding around the temporal layers,
    two-phase GroupNorm with a real cross-rank reduction, distributed backward); (4, 2): both at once -- the layout in which
    the reduced U-Net input gradient once had rank-dependent strides (the un-padded frame shard kept the U-Net's channels-last
    layout) and the CFG-pair all-reduce summed mismatched elements.  f16 on the device: the result differs from the
    single-process one by f16 rounding of a different summation order only: 1e-2 of the tensor's max; the guidance term is
    non-zero."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q, cfg)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert "exception" not in r[1], r[1]["exception"]
    for rank, errs, moved, _ in res:
        assert moved > 1e-4, "guidance did not move the step"
        assert errs["plain"] < 1e-2 and errs["guided"] < 1e-2, (rank, errs)
