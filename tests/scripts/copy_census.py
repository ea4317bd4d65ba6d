"""Who calls the copying tensor ops during one guided DDIM step?  (dev tool)  Patches Tensor.contiguous / clone / copy_ / to / float / half
and torch.cat / zeros_like / zeros / empty_like-free ops with counters keyed by the first caller frame inside this repository; only calls
that really copy (non-contiguous input, dtype change) are counted for contiguous / to.  usage: copy_census.py [height width]"""
import collections
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "guidedvd-3dgs_amd"))
import torch

agg = collections.defaultdict(lambda: [0, 0])
ON = [False]


def caller():
    for fr in reversed(traceback.extract_stack(limit=12)[:-2]):
        if ROOT in fr.filename and "copy_census" not in fr.filename:
            return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
    return "?"


def wrap(owner, name, copies):
    orig = getattr(owner, name)

    def f(*a, **k):
        if ON[0]:
            t = a[0] if a and torch.is_tensor(a[0]) else None
            try:
                does = copies(t, a, k)
            except Exception:
                does = True
            if does:
                r = agg[(name, caller())]
                r[0] += 1
                r[1] += (t.numel() * t.element_size()) if t is not None else 0
        return orig(*a, **k)
    setattr(owner, name, f)


wrap(torch.Tensor, "contiguous", lambda t, a, k: not t.is_contiguous(**k))
wrap(torch.Tensor, "clone", lambda t, a, k: True)
wrap(torch.Tensor, "copy_", lambda t, a, k: True)
wrap(torch.Tensor, "float", lambda t, a, k: t.dtype != torch.float32)
wrap(torch.Tensor, "half", lambda t, a, k: t.dtype != torch.float16)
wrap(torch.Tensor, "to", lambda t, a, k: True)
wrap(torch, "cat", lambda t, a, k: True)
wrap(torch, "stack", lambda t, a, k: True)
wrap(torch, "zeros_like", lambda t, a, k: True)
wrap(torch, "zeros", lambda t, a, k: True)

import bench  # noqa: E402
import argparse  # noqa: E402
hh, ww = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 448)
sys.argv = ["bench.py", "--workload", "ddim_guided", "--ddim-height", str(hh), "--ddim-width", str(ww), "--steps", "1", "--warmup", "2", "--no-cpu-baseline"]
orig_run = bench.ddim_run


def run(args, dev, rank, world, guided, steps, warm, cpu_leg_wanted, cache=None, instrument=True):
    import time
    real_sync = torch.cuda.synchronize
    state = {"n": 0}

    def sync(*a, **k):   # the first synchronize after the warm-up opens the window, the next one closes it
        real_sync(*a, **k)
        state["n"] += 1
        ON[0] = state["n"] == 1
    torch.cuda.synchronize = sync
    try:
        return orig_run(args, dev, rank, world, guided, steps, warm, cpu_leg_wanted, cache, False)
    finally:
        torch.cuda.synchronize = real_sync


bench.ddim_run = run
bench.main()
print("-- copying calls of ONE guided step, by (op, first frame in the repository): count, MB")
for (name, where), (n, b) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{n:5d} {b / 1e6:10.1f} MB  {name:12s} {where}")
