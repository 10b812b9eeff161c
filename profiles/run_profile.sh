# The committed measurements of a round: bench lines (affine = default, linear, single-end, config 5) and the rocprofv3 passes of the
# default command and of the heavy-tailed leg (kernel statistics; FETCH_SIZE and WRITE_SIZE in separate --pmc passes with the kernel trace only).
# usage (on an MI355X box, from the repository root): bash profiles/run_profile.sh [tag]      outputs: gpurun_out/profiles/<tag>_*
set -x
TAG=${1:-r05}
mkdir -p gpurun_out/profiles
timeout 1800 python bench.py > gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.log 2> gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.err; tail -1 gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.log | cut -c1-400
timeout 600 python bench.py --personality linear --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 --steps 5 > gpurun_out/profiles/${TAG}_bench_mapping_pe_linear.log 2>&1
timeout 600 python bench.py --layout se --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 --steps 5 > gpurun_out/profiles/${TAG}_bench_mapping_se_affine.log 2>&1
timeout 1200 python bench.py --read-len 250 --corridor 80 --layout se --subs 0.12 --indel-bases 0.03 --sensitive --workers 4 --steps 3 --no-end-to-end --no-cpu-baseline --heavy-tail-mbp 0 > gpurun_out/profiles/${TAG}_bench_config5_250bp_se_c80_sensitive.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 > $R/gpurun_out/prof_stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats_lin -o stats -- python $R/bench.py --personality linear --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 > $R/gpurun_out/prof_stats_lin.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 > $R/gpurun_out/prof_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 > $R/gpurun_out/prof_write.log 2>&1
# round 5: the heavy-tailed leg alone (GRCh38-like genome of 3.1 Gbp, both sub-legs), the same three passes
timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_hstats -o stats -- python $R/profiles/tools/heavy_leg_only.py --steps 2 --no-cpu-baseline > $R/gpurun_out/prof_hstats.log 2>&1
timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_hfetch -o fetch -- python $R/profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > $R/gpurun_out/prof_hfetch.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_hwrite -o write -- python $R/profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline > $R/gpurun_out/prof_hwrite.log 2>&1
cd $R
S=$(find gpurun_out/prof_stats -name "*.db" | head -1); L=$(find gpurun_out/prof_stats_lin -name "*.db" | head -1); F=$(find gpurun_out/prof_fetch -name "*.db" | head -1); W=$(find gpurun_out/prof_write -name "*.db" | head -1)
HS=$(find gpurun_out/prof_hstats -name "*.db" | head -1); HF=$(find gpurun_out/prof_hfetch -name "*.db" | head -1); HW=$(find gpurun_out/prof_hwrite -name "*.db" | head -1)
python - <<PY
import subprocess, sys
src = open("profiles/summarize_rocprof.py").read().replace('HERE = os.path.dirname(os.path.abspath(__file__))', 'HERE = "gpurun_out/profiles"')
open("gpurun_out/summ.py", "w").write(src)
subprocess.run([sys.executable, "gpurun_out/summ.py", "${TAG}_mapping_pe_affine", "$S", "$F", "$W"])
subprocess.run([sys.executable, "gpurun_out/summ.py", "${TAG}_mapping_pe_linear", "$L"])
subprocess.run([sys.executable, "gpurun_out/summ.py", "${TAG}_heavy_tail", "$HS", "$HF", "$HW"])
PY
# the heavy-tailed genome: per-pass timing of the candidate search (1 Gbp probe, as in round 4), the parity tests' log, parity at scale
NGM_HIP_CS_PHASES=1 timeout 600 python profiles/tools/heavy_tail_probe.py > gpurun_out/profiles/${TAG}_heavy_tail_cs_passes.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_humanlike.py -m gpu -q -s > gpurun_out/profiles/${TAG}_humanlike_parity.log 2>&1
timeout 2400 python profiles/tools/humanlike_t1.py --reads 2000000 > gpurun_out/profiles/${TAG}_humanlike_t1_2M_reads.log 2>&1
python profiles/tools/cpu_scale_probe.py > gpurun_out/profiles/${TAG}_cpu_quota_probe.txt 2>&1
python profiles/tools/kernel_resources.py nextgenmap_amd/build/mapper.o "cs_canon_kernel<3, 6, 2, 1, 7, true>" cs_heavy2 cs_order_kernel cs_order_bucket pair_choice cs_global > gpurun_out/profiles/${TAG}_kernel_registers_and_spills.txt 2>&1
ls -la gpurun_out/profiles
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_lin gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_hstats gpurun_out/prof_hfetch gpurun_out/prof_hwrite
