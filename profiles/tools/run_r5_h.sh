#!/bin/bash
# round 5, call H: heavy2 second pass + coarse 32; padded canonical buckets (headline kernel); rccl test; parity
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline --heavy-tail-mbp 0 > gpurun_out/r5h_bench_main.log 2> gpurun_out/r5h_bench_main.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r5h_bench_main.log') if x.startswith('{')]
d=json.loads(l[-1]); print('main value %.4g ms/step %.2f'%(d['value'],d['ms_per_step']), {k:round(v,2) for k,v in d['kernel_ms'].items()}, 'roofline', round(d['roofline']['frac'],4), 'iso', d['roofline']['candidate_search']['isolated'])
PY
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_dropin.py tests/test_gpu_humanlike.py -x -q -k "candidate_search or rccl or shards or index_files or paired_end_on or other_selection" > gpurun_out/r5h_tests.log 2>&1
tail -5 gpurun_out/r5h_tests.log
NGM_HIP_HOST_TIMING=1 NGM_HIP_CS_PHASES=1 timeout 900 python profiles/tools/heavy_leg_only.py --steps 2 --no-cpu-baseline > gpurun_out/r5h_heavy_leg_timing.json 2> gpurun_out/r5h_heavy_leg_timing.err
grep "heavy class\|pass 1b\|pass 1 \|pass 3\|order replay:" gpurun_out/r5h_heavy_leg_timing.err | sed -n 30,44p | cut -c1-480
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h_heavy_leg_timing.json'))
for leg in ('reads_drawn_uniformly','half_of_the_reads_from_repeats'):
    x=d[leg]; print(leg,'%.3g reads/s'%x['value'],'ms/step %.0f'%x['ms_per_step'],{k:round(v,1) for k,v in x['kernel_ms'].items()}, x['gpu_kernels_fraction_of_step'], x['share_of_reads'])
PY
