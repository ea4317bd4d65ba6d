# run on the GPU box after r5_ab_raster_lib.sh: alternates the two libraries, prints value / sustained / per-kernel us
N=${1:-3}
for i in $(seq $N); do for L in ab new; do
  if [ $L = ab ]; then export GVD_RASTER_LIB=$PWD/guidedvd-3dgs_amd/lib/libgvd_raster_ab.so; else unset GVD_RASTER_LIB; fi
  python bench.py --workload raster --no-cpu-baseline --steps ${STEPS:-200} --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels_us']
print('$L', d['value'], d['sustained']['value'] if d.get('sustained') else None, {a:(round(b,1) if isinstance(b,(int,float)) else b) for a,b in k.items()} if isinstance(k,dict) else k)"
done; done
