# round 3: VALU / LDS / scalar instructions of the candidate search per phase, by difference: the kernel leaves every read after phase 1 (setup:
# codes, k-mers, first lines, chunk items), 2 (+ votes), 3 (+ completion) or not at all (0)
R=$PWD
OUT=$R/gpurun_out/r3_stop
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in 1 2 3 0; do
NGM_HIP_CS_STOP=$s timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/s$s -o s$s -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --workers 1 > $OUT/s$s.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r3/cs_instructions_per_phase.txt
import sqlite3, glob
for s in (1, 2, 3, 0):
    for db in glob.glob("gpurun_out/r3_stop/s%d/*.db" % s) + glob.glob("gpurun_out/r3_stop/s%d/*/*.db" % s):
        c = sqlite3.connect(db).cursor()
        for r in c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like '%cs_canon_kernel%' group by kernel_name, counter_name"):
            print("stop", s, r[1], "%.5g" % r[2], "n", r[3], "dur_us %.1f" % (r[4] / 1000.0))
PY
rm -rf gpurun_out/r3_stop
