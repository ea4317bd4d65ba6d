# quick SQ counter pass over the raster kernels (decision aid; the committed evidence comes from run_raster_prof_all.sh)
R=${GRAFT_REPO_ROOT:-$PWD}
O=/tmp/rprof_sq
rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -- python $R/tests/profile_raster.py 6 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = sorted(glob.glob("$O/pmc_sq/**/*counter_collection.csv", recursive=True))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, c in agg.items():
    m = len(n[k]); d = dur[k] / m
    if d < 3000: continue
    simd_q = 1024 * d * 2.4 / 4
    print(f"{k:42s} n={m} dur={d/1e3:7.1f}us valu_busy={c['SQ_ACTIVE_INST_VALU']/m/simd_q:.3f} insts_valu={c['SQ_INSTS_VALU']/m/1e6:.2f}M wave_cyc={c['SQ_WAVE_CYCLES']/m/1e6:.1f}M "
          f"active_any={c['SQ_ACTIVE_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f} wait_any={c['SQ_WAIT_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f} wait_inst={c['SQ_WAIT_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.2f} occupancy(waves/simd)={c['SQ_WAVE_CYCLES']/m/simd_q:.2f} lds_insts={c['SQ_INSTS_LDS']/m/1e6:.2f}M")
PY
