R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/q_fetch
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/q_fetch -- python $R/tests/profile_raster.py 12 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/q_fetch/**/*counter_collection.csv", recursive=True)[0]
tot, n = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-24:]
    tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in tot: print(k, round(tot[k] / n[k] / 1024, 1), "MB fetch / launch")
PY
