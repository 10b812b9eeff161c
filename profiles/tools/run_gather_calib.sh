# gather calibration on the GPU box: timings + FETCH_SIZE / TCC_EA0_RDREQ per dispatch (separate --pmc passes, kernel-trace only)
R=$PWD
mkdir -p gpurun_out/gather
hipcc --offload-arch=gfx950 -O3 profiles/tools/gather_calib.hip -o gpurun_out/gather/gather_calib || exit 1
gpurun_out/gather/gather_calib 8 | tee gpurun_out/gather/timing.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/gather/p1 -o p -- $R/gpurun_out/gather/gather_calib 8 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $R/gpurun_out/gather/p2 -o p -- $R/gpurun_out/gather/gather_calib 8 > /dev/null 2>&1
cd $R
python - <<'PY' | tee gpurun_out/gather/counters.txt
import sqlite3, glob
for db in sorted(glob.glob("gpurun_out/gather/p*/*.db")):
    c = sqlite3.connect(db).cursor()
    try:
        rows = list(c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection order by dispatch_id"))
    except Exception as e:
        print(db, "ERR", e); continue
    for r in rows: print(r[0].split("(")[0][-48:], r[1], "%.6g" % r[2], r[3])
PY
rm -rf gpurun_out/gather/p1 gpurun_out/gather/p2 gpurun_out/gather/gather_calib
