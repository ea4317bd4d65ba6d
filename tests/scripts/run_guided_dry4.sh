# 4-rank (cfg2 x frames2) guided DDIM dry run on one GPU over gloo, repeated with every distributed hand-off checked
export GVD_DIST_BACKEND=gloo
for i in 1 2 3; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 2965$i tests/scripts/dist_guided_probe.py --workload ddim_guided --ddim-height 192 --ddim-width 256 --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/probe_$i.out 2> gpurun_out/probe_$i.err
  echo "run $i: $(grep -c 'NON-FINITE\|NOT REPLICATED' gpurun_out/probe_$i.out) flagged; $(grep -c AssertionError gpurun_out/probe_$i.err) assertion lines"; grep 'NON-FINITE\|NOT REPLICATED' gpurun_out/probe_$i.out | head -6
done
