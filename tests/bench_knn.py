"""Timing of the simple-knn replacement on the C2-size cloud (GPU) with the C oracle beside it on a bounded sample."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "guidedvd-3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

from simple_knn._C import distCUDA2
from test_knn import _cloud

for P in (200_000, 1_000_000):
    tp = torch.tensor(_cloud("room", P, 11), device="cuda:0")
    for _ in range(3):
        distCUDA2(tp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        distCUDA2(tp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"P={P:8d}  {dt * 1e3:8.3f} ms  {P / dt / 1e6:8.2f} Mpoints/s")
