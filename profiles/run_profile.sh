set -x
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_affine.log 2>&1; tail -1 gpurun_out/bench_affine.log | cut -c1-400
timeout 600 python bench.py --personality linear --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/bench_linear.log 2>&1
timeout 600 python bench.py --layout se --no-cpu-baseline --no-end-to-end --steps 5 > gpurun_out/bench_se.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $R/gpurun_out/prof_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $R/gpurun_out/prof_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $R/gpurun_out/prof_write.log 2>&1
cd $R
find gpurun_out -name "*.db" | head
S=$(find gpurun_out/prof_stats -name "*.db" | head -1); F=$(find gpurun_out/prof_fetch -name "*.db" | head -1); W=$(find gpurun_out/prof_write -name "*.db" | head -1)
cp profiles/summarize_rocprof.py gpurun_out/summ.py
python - <<PY
import shutil,subprocess,sys,os
os.makedirs("gpurun_out/profiles", exist_ok=True)
src=open("profiles/summarize_rocprof.py").read().replace('HERE = os.path.dirname(os.path.abspath(__file__))','HERE = "gpurun_out/profiles"')
open("gpurun_out/summ.py","w").write(src)
subprocess.run([sys.executable,"gpurun_out/summ.py","r02_mapping_pe_affine","$S","$F","$W"])
PY
ls -la gpurun_out/profiles
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
