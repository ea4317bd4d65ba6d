# Multi-rank dry runs on one GPU over gloo: 8-rank (cfg 2 x frames 4) guided + unguided with the hand-off probe, 3-rank (frames 3,
# no CFG pair) guided, and the driver's default line with 4 ranks (raster pairs + ddim).  Small video so that gloo's host staging stays short.
export GVD_DIST_BACKEND=gloo
probe() { timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 tests/scripts/dist_guided_probe.py --workload $3 --ddim-height 192 --ddim-width 256 --gpus $1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/dry8_$2.out 2> gpurun_out/dry8_$2.err; echo "$1 ranks $3: $(grep -c 'NON-FINITE\|NOT REPLICATED' gpurun_out/dry8_$2.out) flagged, $(grep -c 'Error' gpurun_out/dry8_$2.err) error lines; $(grep -o '"value": [0-9.]*' gpurun_out/dry8_$2.out | head -1) $(grep -o '"parallelism": "[^"]*"' gpurun_out/dry8_$2.out | head -1)"; grep 'NON-FINITE\|NOT REPLICATED' gpurun_out/dry8_$2.out | head -3; }
probe 8 29701 ddim_guided
probe 8 29702 ddim
probe 4 29706 ddim
probe 2 29707 ddim
probe 3 29703 ddim_guided
probe 6 29704 ddim_guided
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29705 bench.py --gpus 4 --steps 60 --warmup 10 --ddim-height 192 --ddim-width 256 --no-cpu-baseline > gpurun_out/dry8_all4.out 2> gpurun_out/dry8_all4.err; echo "default line, 4 ranks: $(grep -c Error gpurun_out/dry8_all4.err) error lines"; cut -c1-260 gpurun_out/dry8_all4.out | tail -1
