# The committed measurements of a round: bench lines (affine = default, linear, single-end, config 5), the rocprofv3 passes of the default
# command and of the heavy-tailed leg (kernel statistics; FETCH_SIZE and WRITE_SIZE in separate --pmc passes with the kernel trace only; round 6:
# the heavy leg's PMC passes per sub-leg, summed over the candidate-search group), SQ counters of the search kernel.
# usage (on an MI355X box, from the repository root): bash profiles/run_profile.sh [tag] [part...]     outputs: gpurun_out/profiles/<tag>_*
#   parts: bench lines main heavy sq misc (default: all)
set -x
TAG=${1:-r06}
shift
PARTS=${*:-bench lines main heavy sq misc}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
mkdir -p gpurun_out/profiles
R=$PWD
summ() { python - "$@" <<PY
import subprocess, sys
src = open("profiles/summarize_rocprof.py").read().replace('HERE = os.path.dirname(os.path.abspath(__file__))', 'HERE = "gpurun_out/profiles"')
open("gpurun_out/summ.py", "w").write(src)
subprocess.run([sys.executable, "gpurun_out/summ.py"] + sys.argv[1:])
PY
}
db() { find $1 -name "*.db" | head -1; }
if has bench; then
timeout 2400 python bench.py > gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.log 2> gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.err; tail -1 gpurun_out/profiles/${TAG}_bench_mapping_pe_affine.log | cut -c1-400
fi
if has lines; then
timeout 600 python bench.py --personality linear --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 --steps 5 > gpurun_out/profiles/${TAG}_bench_mapping_pe_linear.log 2>&1
timeout 600 python bench.py --layout se --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 --steps 5 > gpurun_out/profiles/${TAG}_bench_mapping_se_affine.log 2>&1
timeout 1200 python bench.py --read-len 250 --corridor 80 --layout se --subs 0.12 --indel-bases 0.03 --sensitive --workers 4 --steps 3 --no-end-to-end --no-cpu-baseline --heavy-tail-mbp 0 > gpurun_out/profiles/${TAG}_bench_config5_250bp_se_c80_sensitive.log 2>&1
fi
cd /tmp && export TMPDIR=/tmp
if has main; then
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 > $R/gpurun_out/prof_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 > $R/gpurun_out/prof_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 > $R/gpurun_out/prof_write.log 2>&1
(cd $R && summ ${TAG}_mapping_pe_affine "$(db gpurun_out/prof_stats)" "$(db gpurun_out/prof_fetch)" "$(db gpurun_out/prof_write)")
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
fi
if has heavy; then
# the heavy-tailed leg alone (GRCh38-like genome of 3.1 Gbp): kernel statistics over both sub-legs, the PMC passes per sub-leg
timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_hstats -o stats -- python $R/profiles/tools/heavy_leg_only.py --steps 2 --no-cpu-baseline > $R/gpurun_out/prof_hstats.log 2>&1
(cd $R && summ ${TAG}_heavy_tail "$(db gpurun_out/prof_hstats)")
rm -rf $R/gpurun_out/prof_hstats
for sub in uniform repeats; do
timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_hfetch -o fetch -- python $R/profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline --only $sub > $R/gpurun_out/prof_hfetch_$sub.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_hwrite -o write -- python $R/profiles/tools/heavy_leg_only.py --steps 1 --no-cpu-baseline --only $sub > $R/gpurun_out/prof_hwrite_$sub.log 2>&1
(cd $R && summ ${TAG}_heavy_tail_$sub - "$(db gpurun_out/prof_hfetch)" "$(db gpurun_out/prof_hwrite)")
rm -rf $R/gpurun_out/prof_hfetch $R/gpurun_out/prof_hwrite
done
fi
if has sq; then
# SQ counters of the search kernel (one instance, one launch of 1 048 576 reads): VALU / SALU / LDS instructions issued, busy and wait cycles
GROUPS_=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE")
for i in 1 2 3; do
  grp=${GROUPS_[$((i-1))]}
  timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/prof_sq_g$i -o g$i -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --heavy-tail-mbp 0 --workers 1 > $R/gpurun_out/prof_sq_g$i.log 2>&1
done
(cd $R && python - <<'PY' > gpurun_out/profiles/${TAG}_sq_counters_search_and_dp_kernels.txt
import sqlite3, glob
want = ("cs_canon_kernel", "sw_affine_score_pk_kernel", "sw_affine_align_pk_kernel", "gather_pairs_kernel", "select_top1_kernel", "compact_candidates_kernel")
for db in sorted(glob.glob("gpurun_out/prof_sq_g*/*.db") + glob.glob("gpurun_out/prof_sq_g*/*/*.db")):
    c = sqlite3.connect(db).cursor()
    tag = db.split("/")[1]
    try:
        rows = list(c.execute("select kernel_name, grid_size, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, grid_size, counter_name"))
    except Exception as e:
        rows = []
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")
        if any(w in name for w in want): print(tag, name[:70], "grid", r[1], r[2], "%.5g" % r[3], "n", r[4], "dur_us %.1f" % (r[5] / 1000.0 if r[5] else 0))
PY
)
rm -rf $R/gpurun_out/prof_sq_g1 $R/gpurun_out/prof_sq_g2 $R/gpurun_out/prof_sq_g3
fi
cd $R
if has misc; then
# the heavy-tailed genome: per-pass timing of the candidate search (1 Gbp probe, as in rounds 4-5), registers of the kernels
NGM_HIP_CS_PHASES=1 timeout 600 python profiles/tools/heavy_tail_probe.py > gpurun_out/profiles/${TAG}_heavy_tail_cs_passes.txt 2>&1
python profiles/tools/cpu_scale_probe.py > gpurun_out/profiles/${TAG}_cpu_quota_probe.txt 2>&1
python profiles/tools/kernel_resources.py nextgenmap_amd/build/mapper_search.o "cs_canon_kernel<3, 6, 2, 1, 7, true>" cs_heavy2 cs_order_kernel cs_order_bucket cs_global cs_heavy_classify cs_global_prepare compact_candidates > gpurun_out/profiles/${TAG}_kernel_registers_and_spills.txt 2>&1
python profiles/tools/kernel_resources.py nextgenmap_amd/build/mapper.o pair_choice select_top1 gather_pairs >> gpurun_out/profiles/${TAG}_kernel_registers_and_spills.txt 2>&1
python profiles/tools/kernel_resources.py nextgenmap_amd/build/ngm_hip.o "sw_affine_score_pk_kernel<28, false>" "sw_affine_score_pk_kernel<81, false>" sw_affine_score_pk_split "sw_affine_align_pk_kernel<28" >> gpurun_out/profiles/${TAG}_kernel_registers_and_spills.txt 2>&1
fi
ls -la gpurun_out/profiles
