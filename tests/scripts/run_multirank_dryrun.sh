# Dry run of the driver's multi-GPU launch line with 2 ranks on the single GPU of the test box (gloo instead of RCCL).
export GVD_DIST_BACKEND=gloo MIOPEN_USER_DB_PATH=$PWD/guidedvd-3dgs_amd/lvdm_amd/miopen_db
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 100 --warmup 10 2> gpurun_out/dry_raster.err | tail -1 | cut -c1-330
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --workload ddim --ddim-height 320 --ddim-width 448 --gpus 2 --steps 3 --warmup 2 2> gpurun_out/dry_ddim.err | tail -1 | cut -c1-420
for f in gpurun_out/dry_raster.err gpurun_out/dry_ddim.err; do grep -v amdgpu $f | tail -n 2; done
