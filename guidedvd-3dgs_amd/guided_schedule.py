"""Raster group || diffusion group: the schedule of the guidedvd loop over several GPUs (BASELINE configs[3] and [4]).

Reference: train_guidedvd.py places the ViewCrafter model on `cuda:{guidance_gpu_id}` and the 3DGS optimisation on `cuda:0`
(:83,101) and BLOCKS: every `guidance_vd_iter` (260) iterations it renders the 25 guidance views (:500-527), runs the guided
video diffusion (:549-554) and pushes the 24 generated frames onto the pseudo-view stack (:612-636), while the other GPU idles.

Here the two halves are process groups of one `torch.distributed` world (one process per GPU, RCCL):

    raster group     replicas of the Gaussians.  The training step is NOT exchanged: the rasterizer's forward/backward is
                     bitwise deterministic (no float atomics), so replicas that run the same step stay
                     bit-identical with no gradient all-reduce -- cheaper than the 49.6 MB all-reduce per 0.33 ms step.  What
                     IS sharded is the per-view work without gradients: the 25 guidance renders of a diffusion run go per
                     camera over the group (multiview.render_views_sharded, one fused all-gather).
    diffusion group  lvdm_amd.parallel.ParallelPlan over the group's ranks (CFG pair x frame shards).
    hand-off         raster leader -> diffusion ranks: ONE flat fp32 message (guidance images [n,3,H,W], masks [n,1,H,W], raw
                     depths [n,1,H,W], conditioning renders [n,Hc,Wc,3], 2 floats of metadata): 25 x 5 x 480 x 640 x 4 B = 154 MB
                     + 25 x 320 x 448 x 3 x 4 B = 43 MB at the repo-default sizes.  diffusion leader -> raster ranks: ONE
                     message, the generated frames [n,3,h,w] fp32 (43 MB at 320x448, 177 MB at 576x1024).  One large message
                     per peer suits point-to-point xGMI.
    overlap          `deliver_after = D` iterations: the raster group keeps training while the diffusion group works; the
                     frames of the run triggered at iteration i enter the pseudo-view stack at the end of iteration i + D (the
                     raster group blocks there if they are not in yet).  D = 0 is the reference's blocking schedule.  The
                     semantics -- which frames the optimisation sees at which iteration -- depend on D only, not on the
                     layout: one process with both roles, two ranks, or 4 + 4 ranks produce the same sequence
                     (tests/test_guided_schedule*.py), so D is a training hyper-parameter, not a race.

A rank may hold both roles (co-resident: BASELINE configs[3] first half; or "every rank diffuses, raster replicated", which on
MI355X is the faster use of 2 / 8 GPUs because the diffusion run, not the 260 raster iterations, dominates the round).
"""
import time

import torch
import torch.distributed as dist


def _dist_on():
    return dist.is_available() and dist.is_initialized()


class Roles:
    """Which global ranks rasterize and which diffuse.  COLLECTIVE over the default group (every process constructs it with
    the same arguments: `dist.new_group` rule)."""

    def __init__(self, raster_ranks, diffusion_ranks):
        self.raster_ranks = [int(r) for r in raster_ranks]
        self.diffusion_ranks = [int(r) for r in diffusion_ranks]
        if not self.raster_ranks or not self.diffusion_ranks:
            raise ValueError("Roles: both groups need at least one rank")
        on = _dist_on()
        self.world = dist.get_world_size() if on else 1
        self.rank = dist.get_rank() if on else 0
        for name, rk in (("raster", self.raster_ranks), ("diffusion", self.diffusion_ranks)):
            if len(set(rk)) != len(rk) or min(rk) < 0 or max(rk) >= self.world:
                raise ValueError(f"Roles: bad {name} rank list {rk} for a world of {self.world}")
        self.is_raster = self.rank in self.raster_ranks
        self.is_diffusion = self.rank in self.diffusion_ranks
        self.raster_leader, self.diffusion_leader = self.raster_ranks[0], self.diffusion_ranks[0]
        # bridge A: the packet goes from the raster leader to every diffusion rank that has not rendered it itself;
        # bridge B: the frames go from the diffusion leader to every raster rank that has not generated them itself
        self.bridge_a = [self.raster_leader] + [d for d in self.diffusion_ranks if d not in self.raster_ranks]
        self.bridge_b = [self.diffusion_leader] + [r for r in self.raster_ranks if r not in self.diffusion_ranks]
        self.raster_group = self.diffusion_world = self.group_a = self.group_b = None
        if on and self.world > 1:
            self.raster_group = dist.new_group(self.raster_ranks)
            self.group_a = dist.new_group(self.bridge_a) if len(self.bridge_a) > 1 else None
            self.group_b = dist.new_group(self.bridge_b) if len(self.bridge_b) > 1 else None

    @classmethod
    def split(cls, layout="disjoint", world=None):
        """disjoint: first half of the ranks rasterize, second half diffuse (2 GPUs: 1 + 1 = BASELINE configs[3] second half;
        8 GPUs: 4 + 4 = configs[4]).  shared: every rank holds both roles.  A world of one is always co-resident."""
        world = (dist.get_world_size() if _dist_on() else 1) if world is None else world
        if world == 1 or layout == "shared":
            return cls(list(range(world)), list(range(world)))
        if layout != "disjoint":
            raise ValueError(f"Roles.split: unknown layout {layout!r}")
        h = world // 2
        return cls(list(range(h)), list(range(h, world)))

    def describe(self):
        return f"raster ranks {self.raster_ranks} | diffusion ranks {self.diffusion_ranks}"


class PacketSpec:
    """Shapes of one hand-off (known up front on both sides: no size handshake on the wire)."""

    def __init__(self, n_views, height, width, cond_height, cond_width):
        self.n, self.H, self.W, self.Hc, self.Wc = int(n_views), int(height), int(width), int(cond_height), int(cond_width)
        n, hw = self.n, self.H * self.W
        self.sizes = [n * 3 * hw, n * hw, n * hw, n * self.Hc * self.Wc * 3, 2]
        self.numel = sum(self.sizes)

    def bytes(self):
        return 4 * self.numel


class GuidancePacket:
    """What the raster side hands to `run_video_diffusion` (train_guidedvd.py:549-554): guidance images in [0,1] [n,3,H,W], masks
    (1 = guide here; the caller forms 1 - (alpha < 0.9), :535,552) [n,1,H,W], raw depths [n,1,H,W], the conditioning renders
    [n,Hc,Wc,3] (the reference's point-cloud renders) and (iteration, view id)."""

    def __init__(self, images, masks, depths, cond, iteration=0, view=0):
        self.images, self.masks, self.depths, self.cond = images, masks, depths, cond
        self.iteration, self.view = int(iteration), int(view)

    def pack(self, spec):
        meta = torch.tensor([float(self.iteration), float(self.view)], dtype=torch.float32, device=self.images.device)
        parts = [self.images, self.masks, self.depths, self.cond]
        flat = torch.cat([p.reshape(-1).float() for p in parts] + [meta])
        if flat.numel() != spec.numel:
            raise ValueError(f"GuidancePacket: {flat.numel()} elements, the spec says {spec.numel}")
        return flat

    @classmethod
    def unpack(cls, flat, spec):
        a = torch.split(flat, spec.sizes)
        n = spec.n
        return cls(a[0].view(n, 3, spec.H, spec.W), a[1].view(n, 1, spec.H, spec.W), a[2].view(n, 1, spec.H, spec.W),
                   a[3].view(n, spec.Hc, spec.Wc, 3), int(a[4][0].item()), int(a[4][1].item()))


def _bcast(t, src, group):
    """Blocking broadcast of one flat tensor (RCCL on the GPUs; gloo stages device tensors through the host itself)."""
    dist.broadcast(t, src=src, group=group)
    return t


class GuidedSchedule:
    """Runs `total_iters` iterations of the loop on this rank according to its role(s).

        raster side    object with  train_step(it) ; render_guidance(it) -> GuidancePacket (collective over roles.raster_group,
                       replicated result) ; consume_video(it0, video [n,3,h,w] fp32, packet)
        diffusion side object with  generate(packet) -> video [n,3,h,w] (collective over the diffusion group, replicated result)

    cadence / end_sample: a run is triggered at the end of iteration `it` when (it - 1) % cadence == 0 and it < end_sample
    (train_guidedvd.py:431).  deliver_after: see the module docstring (0 <= D <= cadence)."""

    def __init__(self, roles, spec, video_shape, device, cadence=260, end_sample=None, deliver_after=0):
        if not 0 <= deliver_after <= cadence:
            raise ValueError("GuidedSchedule: 0 <= deliver_after <= cadence (one diffusion run in flight at a time)")
        self.roles, self.spec, self.video_shape, self.device = roles, spec, tuple(int(v) for v in video_shape), torch.device(device)
        self.cadence, self.end_sample, self.deliver_after = int(cadence), end_sample, int(deliver_after)
        self.times = {"train": 0.0, "render": 0.0, "generate": 0.0, "wait_video": 0.0, "wait_packet": 0.0, "send": 0.0}
        self.events = []   # (kind, iteration): "trigger", "deliver" -- what the tests compare across layouts

    def triggers(self, total_iters):
        end = total_iters + 1 if self.end_sample is None else self.end_sample
        return [it for it in range(1, total_iters + 1) if (it - 1) % self.cadence == 0 and it < end]

    # -- transport ------------------------------------------------------------------------------------------------
    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _timed(self, key, fn, *a):
        self._sync()
        t0 = time.perf_counter()
        out = fn(*a)
        self._sync()
        self.times[key] += time.perf_counter() - t0
        return out

    def _send_packet(self, pkt):
        r = self.roles
        if r.group_a is not None and r.rank == r.raster_leader:
            _bcast(pkt.pack(self.spec).to(self.device), r.raster_leader, r.group_a)

    def _recv_packet(self):
        r = self.roles
        flat = torch.empty(self.spec.numel, dtype=torch.float32, device=self.device)
        _bcast(flat, r.raster_leader, r.group_a)
        return GuidancePacket.unpack(flat, self.spec)

    def _publish_video(self, video):
        r = self.roles
        if r.group_b is not None and r.rank == r.diffusion_leader:
            _bcast(video.to(self.device, torch.float32).contiguous().reshape(-1), r.diffusion_leader, r.group_b)

    def _recv_video(self):
        r = self.roles
        flat = torch.empty(int(torch.tensor(self.video_shape).prod()), dtype=torch.float32, device=self.device)
        _bcast(flat, r.diffusion_leader, r.group_b)
        return flat.view(self.video_shape)

    # -- the loop -------------------------------------------------------------------------------------------------
    def run(self, total_iters, raster=None, diffusion=None):
        r = self.roles
        if r.is_raster and raster is None or r.is_diffusion and diffusion is None:
            raise ValueError("GuidedSchedule.run: this rank's role object(s) are missing")
        trig = set(self.triggers(total_iters))
        if not r.is_raster:   # diffusion-only rank: serve the runs in trigger order
            for it0 in sorted(trig):
                pkt = self._timed("wait_packet", self._recv_packet)
                video = self._timed("generate", diffusion.generate, pkt)
                self.events.append(("generate", it0))
                self._timed("send", self._publish_video, video)
            return self
        pending = []   # [due iteration, trigger iteration, packet, video or None]

        def deliver(upto):
            while pending and pending[0][0] <= upto:
                _, it0, pkt, video = pending.pop(0)
                if video is None:
                    video = self._timed("wait_video", self._recv_video)
                raster.consume_video(it0, video, pkt)
                self.events.append(("deliver", it0))

        for it in range(1, total_iters + 1):
            if self.device.type == "cuda":
                # no device sync per iteration (it would serialise host and device in a 0.3 ms step): the wall time of the
                # training stream is taken by the caller around run(); the phase times cover the phases that synchronise anyway
                raster.train_step(it)
            else:
                self._timed("train", raster.train_step, it)
            deliver(it)          # frames of earlier runs that are due (before a new packet goes out: one run in flight)
            if it in trig:
                pkt = self._timed("render", raster.render_guidance, it)
                self.events.append(("trigger", it))
                self._timed("send", self._send_packet, pkt)
                video = None
                if r.is_diffusion:   # co-resident rank: the run happens here, now; its frames still enter the stack at `due`
                    video = self._timed("generate", diffusion.generate, pkt)
                    self._timed("send", self._publish_video, video)
                pending.append([it + self.deliver_after, it, pkt, video])
                deliver(it)      # D = 0: the reference's blocking schedule
        deliver(total_iters + self.cadence)   # drain: both sides finish every triggered run
        self._sync()
        return self


# ----------------------------------------------------------------------------------------------------------------------
# The two role objects for synthetic scenes: what bench.py --workload config4 and the GPU tests drive.
# ----------------------------------------------------------------------------------------------------------------------
class RasterTrainer:
    """Raster side: the hot path of one train_guidedvd.py iteration (:320-429) on a synthetic scene -- train view (forward,
    0.8 L1 + 0.2 (1 - SSIM), backward) + one pseudo view once generated frames exist (L1 against the frame, :357-372), one Adam
    step over all Gaussian parameters -- plus the 25 no-grad guidance renders of a diffusion run (:500-527) and the pseudo-view
    stack (:612-636).  The reference's random picks are replaced by deterministic ones (train view it % n, pseudo view by a
    fixed stride) so that runs are comparable across layouts."""

    def __init__(self, scene, traj_cameras, device, roles=None, cond_hw=(320, 448), lr=1e-4, lambda_dssim=0.2):
        import numpy as np
        from diff_gaussian_rasterization import GaussianRasterizationSettings
        self.device, self.roles = torch.device(device), roles
        t = lambda a, rg=False: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device, requires_grad=rg)
        self.P = int(scene["means3D"].shape[0])
        self.params = dict(means3D=t(scene["means3D"], True), opacities=t(scene["opacities"], True), scales=t(scene["scales"], True),
                           rotations=t(scene["rotations"], True), shs=t(scene["shs"], True))
        self.means2D = torch.zeros((self.P, 3), device=self.device, requires_grad=True)
        bg, conf = t(scene["bg"]), torch.ones((self.P, 1), device=self.device)
        deg = int(scene["sh_degree"])

        def settings(c):
            return GaussianRasterizationSettings(image_height=c["image_height"], image_width=c["image_width"], tanfovx=c["tanfovx"],
                                                 tanfovy=c["tanfovy"], bg=bg, scale_modifier=1.0, viewmatrix=t(c["viewmatrix"]),
                                                 projmatrix=t(c["projmatrix"]), sh_degree=deg, campos=t(c["campos"]),
                                                 prefiltered=False, debug=False, confidence=conf)
        self.train_cams = [settings(c) for c in scene["cameras"]]
        self.traj_cams = [settings(c) for c in traj_cameras]
        self.H, self.W = scene["cameras"][0]["image_height"], scene["cameras"][0]["image_width"]
        gen = torch.Generator(device=self.device).manual_seed(7)
        self.gts = [torch.rand((3, self.H, self.W), device=self.device, generator=gen) for _ in self.train_cams]
        self.cond_hw = tuple(cond_hw)
        self.lambda_dssim = lambda_dssim
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, eps=1e-15)
        self.pseudo = []          # [(trajectory camera index, pseudo ground truth [3,H,W])]
        self.runs_consumed = []

    def spec(self):
        return PacketSpec(len(self.traj_cams), self.H, self.W, *self.cond_hw)

    def _render(self, cam):
        from diff_gaussian_rasterization import GaussianRasterizer
        p = self.params
        return GaussianRasterizer(cam)(means3D=p["means3D"], means2D=self.means2D, opacities=p["opacities"], shs=p["shs"],
                                      scales=p["scales"], rotations=p["rotations"])

    def train_step(self, it):
        import fused_loss
        self.opt.zero_grad(set_to_none=True)
        self.means2D.grad = None
        v = it % len(self.train_cams)
        color = self._render(self.train_cams[v])[0]
        loss, _ = fused_loss.photometric_loss(color, self.gts[v], self.lambda_dssim)
        loss.backward()
        if self.pseudo:
            cam_i, gt = self.pseudo[(it * 7) % len(self.pseudo)]
            fused_loss.l1_loss(self._render(self.traj_cams[cam_i])[0], gt).backward()
        self.opt.step()

    def render_guidance(self, it):
        import multiview
        import torch.nn.functional as F

        def fn(cam):
            c, _, d, a = self._render(cam)
            return c, d, a
        with torch.no_grad():
            group = None if self.roles is None else self.roles.raster_group
            if group is None:
                out = torch.stack([torch.cat(fn(cam), 0) for cam in self.traj_cams], 0)
            else:
                out = multiview.render_views_sharded(fn, self.traj_cams, group=group)
            images = out[:, :3].clamp(0, 1)
            alpha = out[:, 4:5].clamp(0, 1)
            masks = 1.0 - (alpha < 0.9).float()                       # train_guidedvd.py:535,552
            depths = out[:, 3:4]
            cond = F.interpolate(images, size=self.cond_hw, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        return GuidancePacket(images, masks, depths, cond.contiguous(), iteration=it, view=it % len(self.train_cams))

    def consume_video(self, it0, video, pkt):
        import torch.nn.functional as F
        frames = F.interpolate(video.to(self.device, torch.float32), size=(self.H, self.W), mode="bilinear", align_corners=False)   # :553-555
        self.pseudo = [(i, frames[i]) for i in range(1, frames.shape[0])]   # the first frame is the training view itself (:614-616)
        self.runs_consumed.append(it0)

    def state(self):
        """Copies of every optimised tensor (the tests compare them bit for bit across runs / layouts)."""
        return {k: v.detach().clone() for k, v in self.params.items()}


class GuidedDiffusionRunner:
    """Diffusion side: one guided 25-frame sample per packet (viewcrafter_wrapper.py:550-573 -> ddim_guidance.py) through
    lvdm_amd's guided sampler; `plan` = the ParallelPlan of the diffusion group (None: one rank).  The conditioning is fixed
    (synthetic: CLIP / DUSt3R are outside the path); the guidance tensors come from the packet.  Every draw of a run is
    seeded from (seed, trigger iteration), so a run's result does not depend on the layout that executed it."""

    def __init__(self, model, cond, uncond, noise_shape, video_hw, device, ddim_steps=50, cfg_scale=7.5, guidance_rescale=0.7,
                 eta=1.0, fs=10, plan=None, decode_group=None, seed=123, guided=True):
        from lvdm_amd.guidance import LossGuidance
        from lvdm_amd.samplers import DDIMSampler, DDIMSamplerGuidance
        self.model, self.cond, self.uncond, self.noise_shape = model, cond, uncond, list(noise_shape)
        self.device, self.plan, self.seed = torch.device(device), plan, int(seed)
        self.kw = dict(S=int(ddim_steps), eta=eta, unconditional_guidance_scale=cfg_scale, guidance_rescale=guidance_rescale,
                       timestep_spacing="uniform_trailing")
        self.fs = torch.tensor([fs], dtype=torch.long, device=self.device)
        self.sampler = (DDIMSamplerGuidance if guided else DDIMSampler)(model)
        self.sampler.parallel = plan
        if decode_group and guided:
            self.sampler.decode_group = int(decode_group)
        self.lg = None
        if guided:
            self.lg = LossGuidance(ddim_steps=int(ddim_steps), recur_steps=1, device=str(self.device))
            self.lg.set_hw(*video_hw)
        self.video_hw = tuple(video_hw)
        self.last_latent = None

    def generate(self, pkt):
        dev = self.device
        if self.lg is not None:
            self.lg.set_guidance_images(pkt.images.to(dev))
            self.lg.set_guidance_masks(pkt.masks.to(dev))
            self.lg.set_guidance_depths(pkt.depths.to(dev))
            self.lg.current_train_iter = pkt.iteration
        run_seed = self.seed + 1000003 * pkt.iteration
        if self.plan is not None:
            self.plan.reseed(run_seed)
        else:
            torch.manual_seed(run_seed)
        extra = {} if self.lg is None else {"loss_guidance_fn": self.lg}
        samples, _ = self.sampler.sample(batch_size=self.noise_shape[0], shape=self.noise_shape[1:], conditioning=self.cond,
                                         unconditional_conditioning=self.uncond, fs=self.fs, verbose=False, **self.kw, **extra)
        self.last_latent = samples.detach()
        with torch.no_grad():
            video = self.model.decode_first_stage(samples.detach())           # [1, 3, T, h, w] in [-1, 1]
        video = (torch.clamp(video[0].float(), -1.0, 1.0) + 1.0) / 2.0       # viewcrafter.py:112, viewcrafter_wrapper.py:573
        return video.permute(1, 0, 2, 3).contiguous()                         # [T, 3, h, w]


def synthetic_latent_diffusion(device, unet_config=None, vae_config=None, seed=0, std=0.02):
    """`lvdm_amd.model.LatentDiffusion` with random weights in the product precision on `device`: fp16 token-major U-Net and
    VAE (MFMA convolutions, flash attention), frozen; zero-initialised modules re-randomised (a fresh U-Net is degenerate:
    its output convolution is zero).  No checkpoints exist offline -- bench.py and the GPU tests use this."""
    from lvdm_amd.model import LatentDiffusion
    device = torch.device(device)
    with torch.device(device):
        ld = LatentDiffusion(unet_config, vae_config)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for p_ in ld.model.parameters():
            if float(p_.abs().max()) == 0.0:
                p_.copy_(torch.randn(p_.shape, device=device, generator=g) * std)
    ld = ld.eval().to(device)
    if device.type == "cuda":
        ld.model.diffusion_model.half().to_token_major()
        ld.first_stage_model.half().to_token_major()
        apply_model, decode_core = ld.apply_model, ld.decode_core
        ld.apply_model = lambda x, t, c, **kw: apply_model(x.half(), t, {k: [v.half() for v in vs] for k, vs in c.items()}, **kw)
        ld.decode_core = lambda z, **kw: decode_core(z.half(), **kw)
        # (decode_first_stage / differentiable_decode_first_stage look decode_core up on the instance: they see the wrapper)
    ld.requires_grad_(False)
    return ld
