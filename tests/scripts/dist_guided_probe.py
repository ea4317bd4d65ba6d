"""Debug aid for the multi-rank guided step: runs bench.py's ddim_guided workload (launch under torch.distributed.run) with every
distributed hand-off of the step checked for finiteness and for being replicated where it must be, and reports the FIRST
offender.  usage: python -m torch.distributed.run --nproc-per-node 4 ... tests/scripts/dist_guided_probe.py <bench.py args>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

import bench
from lvdm_amd import parallel

RANK = int(os.environ.get("RANK", "0"))
bad = []


def check(name, t, replicated=False):
    if not torch.is_tensor(t):
        return
    if not bool(torch.isfinite(t).all()):
        n = int((~torch.isfinite(t)).sum())
        msg = f"[rank {RANK}] NON-FINITE {name}: {n} of {t.numel()} entries, shape {tuple(t.shape)}"
        if not bad:
            print(msg, flush=True)
        bad.append(msg)
    if replicated and dist.is_initialized():
        flat = t.detach().double().nan_to_num().contiguous().reshape(-1)     # row-major order whatever the strides are
        w = torch.arange(1, flat.numel() + 1, device=flat.device, dtype=torch.float64) / flat.numel()
        s = (flat.abs() * w).sum().reshape(1)                                 # position-weighted: a permutation changes it
        lo, hi = s.clone(), s.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if float(hi - lo) > 1e-6 * float(hi.abs() + 1):
            print(f"[rank {RANK}] NOT REPLICATED {name}: checksum spread {float(hi - lo):.3e} of {float(hi):.3e}", flush=True)


P = parallel.ParallelPlan
_ecg, _ig, _gwf, _ec = P.eval_cfg_with_graph, P.input_gradient, P.gather_world_frames, P.eval_cfg


def ec(self, model, x, t, cond, uncond, **kw):
    check("x into the U-Net (plain step)", x, True)
    e_c, e_u = _ec(self, model, x, t, cond, uncond, **kw)
    check("e_cond (plain step)", e_c, True)
    check("e_uncond (plain step)", e_u, True)
    return e_c, e_u



def ecg(self, model, x, t, cond, uncond, **kw):
    check("x into the U-Net", x, True)
    e_c, e_u, graphs = _ecg(self, model, x, t, cond, uncond, **kw)
    for i, x_loc, e_loc in graphs:
        check(f"local e (branch {i})", e_loc)
    check("e_cond (gathered)", e_c, True)
    check("e_uncond (gathered)", e_u, True)
    return e_c, e_u, graphs


def ig(self, graphs, g_cond, g_uncond, like):
    check("cotangent of e_cond", g_cond, True)
    check("cotangent of e_uncond", g_uncond, True)
    if os.environ.get("PROBE_VERBOSE"):
        # the pieces of input_gradient, with checksums: local backward, frame gather, CFG-pair reduction
        g = {0: g_cond, 1: g_uncond}
        for i, x_loc, e_loc in graphs:
            (gx_loc,) = torch.autograd.grad(e_loc, x_loc, self.shard.local(g[i], 2).to(e_loc.dtype).contiguous(), retain_graph=True)
            gx = self.shard.gather(gx_loc.float(), 2) if self.F > 1 else gx_loc.float()
            tot = gx.clone()
            if self.cfg == 2:
                dist.all_reduce(tot, group=self.cfg_group)
            print(f"[rank {RANK} cfg {self.cfg_rank} frame {self.frame_rank}] branch {i}: |gx_loc| {float(gx_loc.float().abs().sum()):.6e} "
                  f"|gx gathered| {float(gx.abs().sum()):.6e} |after cfg all-reduce| {float(tot.abs().sum()):.6e}", flush=True)
    if os.environ.get("PROBE_VERBOSE"):
        print(f"[rank {RANK}] like: shape {tuple(like.shape)} strides {like.stride()} contiguous {like.is_contiguous()} dtype {like.dtype}", flush=True)
    out = _ig(self, graphs, g_cond, g_uncond, like)
    if os.environ.get("PROBE_VERBOSE"):
        print(f"[rank {RANK}] input_gradient out: |.| {float(out.abs().sum()):.6e} strides {out.stride()}", flush=True)
    check("U-Net input gradient (reduced)", out, True)
    return out


def gwf(self, g_local, n_frames):
    check("VAE-side latent gradient, my frames", g_local)
    out = _gwf(self, g_local, n_frames)
    check("VAE-side latent gradient, all frames", out, True)
    return out


P.eval_cfg_with_graph, P.input_gradient, P.gather_world_frames, P.eval_cfg = ecg, ig, gwf, ec
try:
    bench.main()
finally:
    if bad:
        print(f"[rank {RANK}] {len(bad)} non-finite hand-offs; first: {bad[0]}", flush=True)
