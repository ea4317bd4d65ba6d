# Where do k_conv_mfma's SQ_LDS_BANK_CONFLICT cycles come from?  Builds the diffusion library with parts of the convolution's LDS
# traffic compiled out (GVD_CONV_DBG bits, csrc/conv_mfma.hip) and reads the two LDS counters of every variant on one L0 shape.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R/guidedvd-3dgs_amd
for d in 0 1 2 4 8 12 16 32 48; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-pass-failed -fno-honor-nans -DGVD_CONV_DBG=$d \
    -o /tmp/libgvd_conv_dbg$d.so csrc/diffusion_kernels.hip csrc/attention_backward.hip csrc/conv_mfma.hip csrc/gemm_mfma.hip 2>/dev/null &
done
wait
cd /tmp && export TMPDIR=/tmp
cat > /tmp/conv_one.py <<PY
import os, sys
sys.path.insert(0, os.path.join("$R", "guidedvd-3dgs_amd"))
import torch, torch.nn as nn
from lvdm_amd import conv as C
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(25, 72, 128, 640, device="cuda", generator=g).half()
m = nn.Conv2d(640, 640, 3, padding=1).cuda().half().requires_grad_(False)
with torch.no_grad():
    for _ in range(3):
        C.fused_conv(x, m)
torch.cuda.synchronize()
PY
echo "variant  conflict_cycles  lds_active_cycles  lds_insts  duration_us" > $R/gpurun_out/r03_conv_lds_hunt.txt
for d in 0 1 2 4 8 12 16 32 48; do
  rm -rf /tmp/hunt$d
  GVD_DIFFUSION_LIB=/tmp/libgvd_conv_dbg$d.so timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d /tmp/hunt$d -- python /tmp/conv_one.py > /dev/null 2>&1
  python - $d >> $R/gpurun_out/r03_conv_lds_hunt.txt <<PY
import csv, glob, sys, collections
d = sys.argv[1]
fs = glob.glob(f"/tmp/hunt{d}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list); dur = []
for r in csv.DictReader(open(fs[0])):
    if "k_conv_mfma" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r.get("End_Timestamp"): dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
m = lambda k: sum(acc[k]) / max(1, len(acc[k]))
print(f"{d:>7}  {m('SQ_LDS_BANK_CONFLICT'):15.0f}  {m('SQ_ACTIVE_INST_LDS'):17.0f}  {m('SQ_INSTS_LDS'):9.0f}  {sum(dur) / max(1, len(dur)):10.1f}")
PY
done
