"""Where do the guided step's strided-copy / cast / add launches come from?  (round 5 dev tool)  Runs bench.ddim_run (guided, 320 x 448, 1 warm-up
+ 2 steps) with torch.Tensor.contiguous / clone / to / float / half / add / mul and torch.cat wrapped: every call that makes a NEW device tensor
of >= 64 K elements is booked under the first stack frame inside lvdm_amd (or bench.py).  Python-level calls only (incl. the backward methods of
the package's autograd Functions, which run on autograd's thread); copies made inside ATen are not seen."""
import collections, os, sys, traceback, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
book = collections.defaultdict(lambda: [0, 0])
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "lvdm_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:110]}"
    return "?"
def wrap(owner, name, label, is_copy):
    fn = getattr(owner, name)
    def w(*a, **k):
        r = fn(*a, **k)
        if isinstance(r, torch.Tensor) and r.is_cuda and r.numel() >= 65536 and is_copy(a, r):
            e = book[(label, site())]; e[0] += 1; e[1] += r.numel() * r.element_size()
        return r
    setattr(owner, name, w)
new_storage = lambda a, r: not (isinstance(a[0], torch.Tensor) and a[0].data_ptr() == r.data_ptr())
for nm in ("contiguous", "clone", "to", "float", "half", "reshape"):
    wrap(torch.Tensor, nm, nm, new_storage)
wrap(torch, "cat", "cat", lambda a, r: True)
for nm in ("__add__", "__mul__", "__sub__", "add", "mul"):
    wrap(torch.Tensor, nm, nm.strip("_"), lambda a, r: True)
sys.argv = ["bench.py", "--workload", "ddim_guided", "--ddim-height", "320", "--ddim-width", "448", "--no-cpu-baseline"]
args = None
ap_main = bench.main
# reuse bench's own argument parser by running main() up to the dispatch: simplest is to call ddim_run with a parsed namespace
import argparse
ns = argparse.Namespace(frames=25, ddim_height=320, ddim_width=448, batch_cfg=False, no_batch_cfg=False, ae_frames=None, graph=False, cpu_seconds=0.0)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
STEPS = 3
bench.ddim_run(ns, dev, 0, 1, guided=True, steps=STEPS - 1, warm=1, cpu_leg_wanted=False, instrument=False)
rows = sorted(book.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"per guided step: {sum(v[0] for _, v in rows) / STEPS:.0f} booked calls, {tot / STEPS / 1e6:.0f} MB written by them")
for (label, where), (n, b) in rows[:40]:
    print(f"{b / STEPS / 1e6:8.1f} MB/step  n={n / STEPS:6.1f}/step  {label:10s} {where}")
