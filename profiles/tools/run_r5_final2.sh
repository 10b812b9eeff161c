#!/bin/bash
# round 5, last call: the bench line and the heavy leg's rocprofv3 passes of the final tree (after cs_heavy2_kernel's restart with more parts; eight instances on the heavy leg)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=r05
mkdir -p gpurun_out/profiles
timeout 1500 python bench.py > gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.log 2> gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.err; tail -1 gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.log | cut -c1-300
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_hstats -o stats -- python $R/profiles/tools/heavy_leg_only.py --steps 2 --no-cpu-baseline > $R/gpurun_out/prof_hstats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_hfetch -o fetch -- python $R/profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > $R/gpurun_out/prof_hfetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_hwrite -o write -- python $R/profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > $R/gpurun_out/prof_hwrite.log 2>&1
cd $R
HS=$(find gpurun_out/prof_hstats -name "*.db" | head -1); HF=$(find gpurun_out/prof_hfetch -name "*.db" | head -1); HW=$(find gpurun_out/prof_hwrite -name "*.db" | head -1)
python - <<PY
import subprocess, sys
src = open("profiles/summarize_rocprof.py").read().replace('HERE = os.path.dirname(os.path.abspath(__file__))', 'HERE = "gpurun_out/profiles"')
open("gpurun_out/summ.py", "w").write(src)
subprocess.run([sys.executable, "gpurun_out/summ.py", "${TAG}_heavy_tail", "$HS", "$HF", "$HW"])
PY
NGM_HIP_CS_PHASES=1 timeout 400 python profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > /dev/null 2> gpurun_out/profiles/${TAG}_heavy_tail_3100mbp_phases.txt
python profiles/tools/kernel_resources.py nextgenmap_amd/build/mapper.o "cs_canon_kernel<3, 6, 2, 1, 7, true>" cs_heavy2 cs_order_kernel cs_order_bucket pair_choice cs_global > gpurun_out/profiles/${TAG}_kernel_registers_and_spills.txt 2>&1
rm -rf gpurun_out/prof_hstats gpurun_out/prof_hfetch gpurun_out/prof_hwrite
ls -la gpurun_out/profiles
