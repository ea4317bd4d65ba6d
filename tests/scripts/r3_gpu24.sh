python -m pytest tests/test_gemm_gpu.py tests/test_diffusion_gpu.py tests/test_guided_schedule.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r3_t24.log
python - > gpurun_out/r3_rowstats.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, 'guidedvd-3dgs_amd')
from lvdm_amd import gemm
for M, C in [(230400, 320), (57600, 640), (14400, 1280), (230400, 512)]:
    x = torch.randn(M, C, device='cuda').half()
    for _ in range(3): gemm.row_stats(x, 1e-5)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): gemm.row_stats(x, 1e-5)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    print(f"row_stats M={M} C={C}: {us:.1f} us  {M * C * 2 / us / 1e6:.2f} TB/s")
PY
python bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_guided_320_auto.json 2>> gpurun_out/r3_guided_320.err
python bench.py --workload ddim_guided --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_guided_576_auto.json 2>> gpurun_out/r3_guided_320.err
python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r3_config4_b.json 2>> gpurun_out/r3_guided_320.err
