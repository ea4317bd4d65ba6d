# rocprofv3 kernel trace of a DDIM bench run; args are passed to bench.py.  Output: gpurun_out/prof_<TAG>/

export GVD_CONV_FIND=1
export GVD_BENCH_MARKERS=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG:-ddim} -- python $R/bench.py --steps ${STEPS:-1} --warmup ${WARMUP:-1} "$@" 2>/dev/null | tail -1 | cut -c1-300
