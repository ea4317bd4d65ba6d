export GVD_DIST_BACKEND=gloo MIOPEN_USER_DB_PATH=$PWD/guidedvd-3dgs_amd/lvdm_amd/miopen_db
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --workload ddim --ddim-height 320 --ddim-width 448 --gpus 4 --steps 2 --warmup 1 2> gpurun_out/dry_ddim4.err | tail -1 | cut -c1-700
grep -v amdgpu gpurun_out/dry_ddim4.err | tail -n 3
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29614 bench.py --workload ddim_guided --ddim-height 320 --ddim-width 448 --gpus 4 --steps 1 --warmup 1 2> gpurun_out/dry_guided4.err | tail -1 | cut -c1-300
grep -v amdgpu gpurun_out/dry_guided4.err | tail -n 3
