import sys,time,torch
sys.path.insert(0,"guidedvd-3dgs_amd"); sys.path.insert(0,"tests")
import numpy as np
import test_knn as t
from simple_knn._C import distCUDA2
x=torch.tensor(t._cloud("room",200_000,11),device="cuda:0")
distCUDA2(x); torch.cuda.synchronize()
t0=time.time()
for _ in range(10): distCUDA2(x)
torch.cuda.synchronize(); print("room 200k", (time.time()-t0)/10*1e3, "ms")
