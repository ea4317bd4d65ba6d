# what the driver does at round end, in one call: build check + smoke() on cuda:0 + the bench contract line
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/contract_line.json 2> gpurun_out/contract.err
