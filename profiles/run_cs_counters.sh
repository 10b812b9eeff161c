# SQ / TCP / TCC counters of the candidate-search kernel (one rocprofv3 --pmc pass per group; kernel-trace only)
# usage: bash profiles/run_cs_counters.sh [group numbers...]
R=$PWD
mkdir -p gpurun_out/cs_pmc
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
         "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE")
SEL=${@:-1 2 3 4}
for i in $SEL; do
  grp=${GROUPS_[$((i-1))]}
  timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/cs_pmc/g$i -o g$i -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --workers 1 > $R/gpurun_out/cs_pmc/g$i.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/cs_pmc/summary.txt
import sqlite3, glob
for db in sorted(glob.glob("gpurun_out/cs_pmc/g*/*.db")):
    c = sqlite3.connect(db).cursor()
    try:
        rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%cs_bucket%' group by kernel_name, counter_name"))
    except Exception as e:
        print(db, "ERR", e); continue
    for r in rows: print(r[0].split("(")[0][-40:], r[1], "%.4g" % r[2], r[3])
PY
rm -rf gpurun_out/cs_pmc/g*/
