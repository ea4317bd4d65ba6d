# exercise the multi-GPU bench code path with the one GPU we have: torch.distributed.run, 1 rank (plan = None, nccl init/barrier)
export MIOPEN_USER_DB_PATH=$PWD/guidedvd-3dgs_amd/lvdm_amd/miopen_db
export GVD_CONV_FIND=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --workload ddim --gpus 1 --steps 3 --warmup 2 2>gpurun_out/dist1.err | tail -1 | cut -c1-1800
