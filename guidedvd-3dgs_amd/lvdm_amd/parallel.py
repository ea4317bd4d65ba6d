"""Multi-GPU decomposition of the DDIM step: CFG pair x frame shards, one process per GPU.

The reference is single-GPU (`DDIMSampler.p_sample_ddim`, ddim.py:208-280, evaluates cond and uncond
sequentially; every layer sees all 25 frames).  The path partitions two ways:

* **CFG pair** (`cfg` group, size 1 or 2): the two `apply_model` calls of a step are independent given x_t, so rank
  c evaluates one of them; one all-gather of e_t (3.7 MB @576x1024) per step.  Guided step: each rank back-propagates
  through its own branch and the 3.7 MB x-gradients are summed with one all-reduce.
* **Frame shards** (`frames` group, size F): spatial layers (ResBlock conv2d, SpatialTransformer, Down/Upsample)
  are per-frame, so rank f keeps frames [t0_f, t1_f) of every feature map.  Temporal layers couple the T frames of
  one pixel: around each TemporalTransformer / TemporalConvBlock the token-major activation is re-sharded
  frames -> pixels with one all-to-all ([T_f, P, C] -> [T, P_f, C]) and back afterwards (Ulysses-style); their 5-D
  GroupNorm statistics (over T, h, w) are completed with a 2*G-double all-reduce (ops.group_norm(group=...)).
  xGMI is point-to-point, and an all-to-all uses all links at once: 147 MB / F per rank at L0.

world = cfg x frames, rank = cfg_rank * F + frame_rank (frame shards of one CFG branch are neighbours).
Everything here is `torch.distributed` (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU tests).
"""
import contextlib

import torch
import torch.distributed as dist


def split_counts(n, parts):
    """n items over `parts` ranks, sizes differing by at most one (25 frames over 4 -> 7, 6, 6, 6)."""
    base, rem = divmod(n, parts)
    return [base + 1 if r < rem else base for r in range(parts)]


class FrameShard:
    """Frame-parallel group: which frames this rank owns, and the all-to-all geometry."""

    def __init__(self, group, n_frames):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.T = n_frames
        self.counts = split_counts(n_frames, self.world)
        self.offsets = [sum(self.counts[:r]) for r in range(self.world)]
        if min(self.counts) < 1:
            raise ValueError(f"{n_frames} frames cannot be split over {self.world} ranks")

    @property
    def lo(self):
        return self.offsets[self.rank]

    @property
    def hi(self):
        return self.offsets[self.rank] + self.counts[self.rank]

    def local(self, x, dim):
        """This rank's frames of a full tensor whose `dim` is the frame axis."""
        return x.narrow(dim, self.lo, self.counts[self.rank])

    def gather(self, x_local, dim):
        """All frames from every rank's local slice (no autograd; sampler-level tensors).  Slices are padded to the
        largest count so the all-gather is equal-sized on every backend."""
        dim = dim % x_local.dim()
        x_local = x_local.contiguous()   # one (row-major) layout on every rank: what comes back is replicated, strides included
        cmax = max(self.counts)
        pad = cmax - x_local.shape[dim]
        if pad:
            shp = list(x_local.shape)
            shp[dim] = pad
            x_local = torch.cat([x_local, x_local.new_zeros(shp)], dim=dim)
        parts = [torch.empty_like(x_local) for _ in range(self.world)]
        dist.all_gather(parts, x_local, group=self.group)
        return torch.cat([p.narrow(dim, 0, c) for p, c in zip(parts, self.counts)], dim=dim)


_ACTIVE = None
_PLANS_BUILT = 0   # ParallelPlan objects constructed by this process (decorrelates their noise streams)


def active():
    """The FrameShard the U-Net is currently running under (None = all frames local)."""
    return _ACTIVE


@contextlib.contextmanager
def frame_parallel(shard):
    global _ACTIVE
    prev, _ACTIVE = _ACTIVE, (shard if shard is not None and shard.world > 1 else None)
    try:
        yield
    finally:
        _ACTIVE = prev


def _a2a(src, n_out, out_splits, in_splits, group):
    """all_to_all_single; a backend without a device all-to-all (gloo, used to run two ranks on one GPU in the tests)
    gets the exchange staged through host memory -- RCCL takes the device tensors directly."""
    src = src.contiguous()
    if src.is_cuda and dist.get_backend(group) == "gloo":
        host = src.cpu()
        out = host.new_empty(n_out)
        dist.all_to_all_single(out, host, out_splits, in_splits, group=group)
        return out.to(src.device)
    out = src.new_empty(n_out)
    dist.all_to_all_single(out, src, out_splits, in_splits, group=group)
    return out


class _AllToAll(torch.autograd.Function):
    """all_to_all_single with explicit split sizes; the backward is the transposed exchange."""

    @staticmethod
    def forward(ctx, flat, in_splits, out_splits, group):
        ctx.cfg = (in_splits, out_splits, group)
        return _a2a(flat, sum(out_splits), out_splits, in_splits, group)

    @staticmethod
    def backward(ctx, g):
        in_splits, out_splits, group = ctx.cfg
        return _a2a(g, sum(in_splits), in_splits, out_splits, group), None, None, None


def frames_to_pixels(tok, shard):
    """[T_f, P, C] (this rank's frames, all pixels) -> [T, P_f, C] (all frames, this rank's pixels)."""
    Tf, P, C = tok.shape
    pc = split_counts(P, shard.world)
    po = [sum(pc[:r]) for r in range(shard.world)]
    send = torch.cat([tok[:, po[r]:po[r] + pc[r]].reshape(-1) for r in range(shard.world)])
    mine = pc[shard.rank]
    out = _AllToAll.apply(send, [Tf * pc[r] * C for r in range(shard.world)],
                          [shard.counts[s] * mine * C for s in range(shard.world)], shard.group)
    return out.view(shard.T, mine, C)   # rank s's block is [T_s, P_f, C]: stacking them IS the frame order


def pixels_to_frames(t, shard, P):
    """[T, P_f, C] -> [T_f, P, C]; inverse of frames_to_pixels (P = total pixels)."""
    T, mine, C = t.shape
    pc = split_counts(P, shard.world)
    Tf = shard.counts[shard.rank]
    recv = _AllToAll.apply(t.reshape(-1), [shard.counts[s] * mine * C for s in range(shard.world)],
                           [Tf * pc[r] * C for r in range(shard.world)], shard.group)
    blocks, off = [], 0
    for r in range(shard.world):
        n = Tf * pc[r] * C
        blocks.append(recv[off:off + n].view(Tf, pc[r], C))
        off += n
    return torch.cat(blocks, dim=1)


def plan_seed(user_seed, n_built):
    """Replicated-noise seed of the `n_built`-th plan of a process whose torch seed is `user_seed` (splitmix64 finaliser over
    the pair): a function of the user's seed only, never equal to it."""
    m = (1 << 64) - 1
    z = (int(user_seed) + (int(n_built) + 1) * 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return (z ^ (z >> 31)) % (1 << 63)


class ParallelPlan:
    """cfg x frames layout over an initialised default process group, or over a SUB-GROUP of it.

    ranks: the global ranks the plan spans, in plan order (default: every rank of the default group).  The constructor is
    COLLECTIVE OVER THE DEFAULT GROUP -- `dist.new_group` must be called by every process, members or not -- so a process that
    is not in `ranks` constructs the plan too and gets `.member == False` (it must not call anything else on it).  This is
    what lets a node be split into a diffusion group and a raster group (guided_schedule.Roles: BASELINE configs[3] / [4])."""

    def __init__(self, n_frames, cfg=None, world=None, rank=None, ranks=None):
        g_world, g_rank = dist.get_world_size(), dist.get_rank()
        if ranks is None:
            ranks = list(range(g_world if world is None else world))
        ranks = [int(r) for r in ranks]
        if len(set(ranks)) != len(ranks) or min(ranks) < 0 or max(ranks) >= g_world:
            raise ValueError(f"ParallelPlan: bad rank list {ranks} for a default group of {g_world}")
        world = len(ranks)
        if rank is None:
            rank = ranks.index(g_rank) if g_rank in ranks else -1
        self.ranks, self.member = ranks, rank >= 0
        if cfg is None:
            cfg = 2 if world % 2 == 0 else 1
        if world % cfg:
            raise ValueError(f"world {world} is not a multiple of the CFG degree {cfg}")
        F = world // cfg
        self.world, self.rank, self.cfg, self.F = world, rank, cfg, F
        self.cfg_rank, self.frame_rank = (rank // F, rank % F) if self.member else (-1, -1)
        # every rank of the default group must create every group, in the same order
        self.world_group = dist.group.WORLD if ranks == list(range(g_world)) else dist.new_group(ranks)
        self.frame_group = self.cfg_group = None
        for c in range(cfg):
            g = dist.new_group([ranks[c * F + f] for f in range(F)])
            if c == self.cfg_rank:
                self.frame_group = g
        for f in range(F):
            g = dist.new_group([ranks[c * F + f] for c in range(cfg)])
            if f == self.frame_rank:
                self.cfg_group = g
        self._world_shards = {}
        self._generators = {}
        global _PLANS_BUILT
        n_built, _PLANS_BUILT = _PLANS_BUILT, _PLANS_BUILT + 1
        if not self.member:
            self.shard, self.seed = None, None
            return
        self.shard = FrameShard(self.frame_group, n_frames)
        # The latent x is replicated: every rank must draw the same x_T / per-step noise whatever its own RNG state is.
        # One seed broadcast from the plan's first rank; the samplers draw from plan.generator(device).  The seed is that
        # rank's torch seed (torch.manual_seed / seed_everything, as the reference's ViewCrafter driver sets it --
        # viewcrafter_wrapper.py:253-262), so user seeding governs the multi-GPU run exactly as it governs the single-GPU one
        # (same seed, same videos; a different seed, different videos).  The seed is passed through a fixed mix, for the first
        # plan too: a generator seeded with initial_seed() itself would REPEAT the device generator's Philox stream from offset
        # 0, so any earlier global draw of the same size (the VAE posterior sample of encode_first_stage, which runs before
        # the sampler) would be value-identical to x_T.  Every plan of a process gets its own stream.  `reseed()` is the
        # explicit override.
        seed = [plan_seed(torch.initial_seed(), n_built) if rank == 0 else 0]
        dist.broadcast_object_list(seed, src=ranks[0], group=self.world_group)
        self.seed = int(seed[0])

    def generator(self, device):
        """The replicated-draw generator for `device` (same stream of numbers on every rank)."""
        device = torch.device(device)
        g = self._generators.get(device)
        if g is None:
            g = self._generators[device] = torch.Generator(device=device).manual_seed(self.seed)
        return g

    def reseed(self, seed):
        """Restart the replicated noise stream with an explicit seed (call with the same value on every rank).  Not needed for
        reproducibility: the default stream already follows rank 0's torch.manual_seed."""
        self.seed = int(seed)
        self._generators = {}

    def check_replicated(self, x, what="latent"):
        """Debug aid: raise if `x` differs across ranks (one tiny all-reduce of a checksum)."""
        s = x.detach().double().sum().reshape(1)
        lo, hi = s.clone(), s.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.world_group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.world_group)
        if float(hi - lo) != 0.0:
            raise RuntimeError(f"{what} is not replicated across ranks (checksum spread {float(hi - lo):.3e})")

    # -- one U-Net evaluation on this rank's frames, result gathered to all frames -----------------------------
    def _shard_cond(self, cond):
        out = dict(cond)
        if "c_concat" in cond:
            out["c_concat"] = [self.shard.local(c, 2).contiguous() for c in cond["c_concat"]]
        return out

    def apply_local(self, model, x, t, cond, **kw):
        """model.apply_model on this rank's frame slice (x [b, C, T, h, w] full, replicated)."""
        x_loc = self.shard.local(x, 2).contiguous()
        with frame_parallel(self.shard):
            return model.apply_model(x_loc, t, self._shard_cond(cond), **kw)

    def eval_cfg(self, model, x, t, cond, uncond, **kw):
        """(e_cond, e_uncond) for the full latent, no autograd: the plain sampler's two apply_model calls."""
        branches = [cond, uncond]
        mine = [self.cfg_rank] if self.cfg == 2 else [0, 1]
        got = {}
        for i in mine:
            e_loc = self.apply_local(model, x, t, branches[i], **kw)
            got[i] = self.shard.gather(e_loc, 2) if self.F > 1 else e_loc
        if self.cfg == 2:
            both = [torch.empty_like(got[self.cfg_rank]) for _ in range(2)]
            dist.all_gather(both, got[self.cfg_rank].contiguous(), group=self.cfg_group)
            return both[0], both[1]
        return got[0], got[1]

    # -- guided step: forward that keeps the local autograd graph, then the distributed x-gradient ---------------
    def eval_cfg_with_graph(self, model, x, t, cond, uncond, **kw):
        """Returns (e_cond, e_uncond) detached full tensors and an opaque handle for `input_gradient`."""
        branches = [cond, uncond]
        mine = [self.cfg_rank] if self.cfg == 2 else [0, 1]
        full, graphs = {}, []
        for i in mine:
            x_loc = self.shard.local(x.detach(), 2).contiguous().requires_grad_(True)
            with frame_parallel(self.shard):
                e_loc = model.apply_model(x_loc, t, self._shard_cond(branches[i]), **kw)
            graphs.append((i, x_loc, e_loc))
            full[i] = self.shard.gather(e_loc.detach(), 2) if self.F > 1 else e_loc.detach()
        if self.cfg == 2:
            both = [torch.empty_like(full[self.cfg_rank]) for _ in range(2)]
            dist.all_gather(both, full[self.cfg_rank].contiguous(), group=self.cfg_group)
            full = {0: both[0], 1: both[1]}
        return full[0], full[1], graphs

    def input_gradient(self, graphs, g_cond, g_uncond, like):
        """sum over branches of J(e_branch -> x)^T g_branch for the FULL x (replicated result).  g_* are the full
        [b, 4, T, h, w] cotangents of e_cond / e_uncond (identical on every rank)."""
        g = {0: g_cond, 1: g_uncond}
        # A fresh ROW-MAJOR buffer, not zeros_like(like): an all-reduce sums storage elements, and `like` (an autograd
        # gradient) can carry a permuted-dense layout on one rank and the standard one on another -- e.g. the rank whose frame
        # slice needed no padding in FrameShard.gather keeps the U-Net's channels-last strides.  Found by the 4-rank
        # (cfg 2 x frames 2) dry run on the device: two of the four ranks summed mismatched elements.
        total = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
        for i, x_loc, e_loc in graphs:
            # every rank of the frame group runs this backward together: the reversed all-to-alls route the
            # cotangents of other ranks' frames through the temporal layers into this rank's x_loc
            (gx_loc,) = torch.autograd.grad(e_loc, x_loc, self.shard.local(g[i], 2).to(e_loc.dtype).contiguous())
            gx = self.shard.gather(gx_loc.float(), 2) if self.F > 1 else gx_loc.float()
            total += gx.to(total.dtype)
        if self.cfg == 2:
            dist.all_reduce(total, group=self.cfg_group)
        return total

    def frame_owner_slices(self, n_frames):
        """Frames whose VAE decode + loss gradient this rank computes in the guided step (all `world` ranks share)."""
        counts = split_counts(n_frames, self.world)
        lo = sum(counts[:self.rank])
        return lo, lo + counts[self.rank], counts

    def gather_world_frames(self, g_local, n_frames):
        """[b, 4, my frames, h, w] per-frame gradients of `frame_owner_slices` -> all frames, on every rank."""
        ws = self._world_shards.get(n_frames)
        if ws is None:
            ws = self._world_shards[n_frames] = FrameShard(self.world_group, n_frames)
        return ws.gather(g_local, 2)
