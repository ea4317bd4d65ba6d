bash tests/scripts/run_multirank_dryrun.sh > gpurun_out/r3_dry2.log 2>&1
export GVD_DIST_BACKEND=gloo
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29705 bench.py --gpus 4 --steps 60 --warmup 10 --ddim-height 192 --ddim-width 256 --ddim-steps 6 --no-cpu-baseline > gpurun_out/r3_dry4.out 2> gpurun_out/r3_dry4.err
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29706 bench.py --workload ddim_guided --ddim-height 192 --ddim-width 256 --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_dry2g.out 2> gpurun_out/r3_dry2g.err
