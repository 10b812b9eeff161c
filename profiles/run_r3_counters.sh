# round 3: SQ counters of the final candidate-search kernel and of the DP kernels.
# One rocprofv3 --pmc pass per counter group, kernel trace only (no other trace domains).  usage: bash profiles/run_r3_counters.sh
R=$PWD
OUT=$R/gpurun_out/r3_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum")
for pers in affine linear; do
for i in 1 2 3 4; do
  grp=${GROUPS_[$((i-1))]}
  timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $OUT/${pers}_g$i -o g$i -- python $R/bench.py --personality $pers --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --workers 1 > $OUT/${pers}_g$i.log 2>&1
done
done
for f in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $f --kernel-trace -d $OUT/affine_$f -o $f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --workers 1 > $OUT/affine_$f.log 2>&1
done
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/affine_stats -o stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/affine_stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/linear_stats -o stats -- python $R/bench.py --personality linear --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/linear_stats.log 2>&1
cd $R
python - <<'PY' | tee gpurun_out/r3_pmc/summary.txt
import sqlite3, glob, os
want = ("cs_canon_kernel", "cs_fast2_kernel", "sw_affine_score_pk_kernel", "sw_affine_align_pk_kernel", "sw_score_pk_kernel", "sw_align_pk_kernel", "sw_align_kernel", "sam_write_kernel", "sam_lengths_kernel", "gather_pairs_kernel")
for db in sorted(glob.glob("gpurun_out/r3_pmc/*/*.db") + glob.glob("gpurun_out/r3_pmc/*/*/*.db")):
    c = sqlite3.connect(db).cursor()
    tag = db.split("/")[2]
    try:
        rows = list(c.execute("select kernel_name, grid_size, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, grid_size, counter_name"))
    except Exception as e:
        rows = []
    try:
        rows2 = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if "stats" in tag:
            for r in rows2: print(tag, "STATS", r[0].split("(")[0].replace("void ", "")[:70], "calls", r[1], "avg_us %.1f" % (r[3] / 1000.0), "pct %.2f" % r[4])
    except Exception as e2:
        pass
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")
        if any(w in name for w in want): print(tag, name[:70], "grid", r[1], r[2], "%.5g" % r[3], "n", r[4], "dur_us %.1f" % (r[5] / 1000.0 if r[5] else 0))
PY
rm -rf gpurun_out/r3_pmc/*/
