R=$PWD
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
G2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
i=0
for grp in "$G1" "$G2"; do i=$((i+1));
  timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/prof_sqh_g$i -o g$i -- python $R/profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline --only uniform --workers 1 > $R/gpurun_out/prof_sqh_g$i.log 2>&1
done
cd $R
python - <<'PY' > gpurun_out/r06_sq_counters_heavy_leg_kernels.txt
import sqlite3, glob
want = ("cs_heavy2_kernel", "cs_order_bucket_kernel", "cs_fast_kernel", "cs_order_kernel")
for db in sorted(glob.glob("gpurun_out/prof_sqh_g*/*.db") + glob.glob("gpurun_out/prof_sqh_g*/*/*.db")):
    c = sqlite3.connect(db).cursor()
    tag = db.split("/")[1]
    rows = list(c.execute("select kernel_name, grid_size, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, grid_size, counter_name"))
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")
        if any(w in name for w in want): print(tag, name[:60], "grid", r[1], r[2], "%.5g" % r[3], "n", r[4], "dur_us %.1f" % (r[5] / 1000.0 if r[5] else 0))
PY
rm -rf gpurun_out/prof_sqh_g1 gpurun_out/prof_sqh_g2
grep "heavy2_kernel<512>" gpurun_out/r06_sq_counters_heavy_leg_kernels.txt
