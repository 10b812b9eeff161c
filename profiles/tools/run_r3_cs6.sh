# round 3: candidate search -- register budget (waves per SIMD the allocation aims at): 7 (72 VGPRs, 11 spilled dwords), 6 (80, 5), 5 (86, 0)
mkdir -p gpurun_out/r3
for w in 7 6 5 7 6 5; do
NGM_HIP_CS_CANON_WPE=$w NGM_HIP_CS_PHASES=1 timeout 900 python bench.py --steps 5 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_cs6_w$w.log 2> gpurun_out/r3/bench_cs6_w$w.err; python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_cs6_w$w.log') if l.startswith('{')][0])
print('wpe $w', j['value'], j['ms_per_step'], j['kernel_ms']['candidate_search'], j['roofline']['frac'])
PY
grep "cs fast\|cs canonical path (shape\|in front" gpurun_out/r3/bench_cs6_w$w.err | tail -3
done
