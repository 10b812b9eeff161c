# round 3: parallel FASTQ record index in ngm-hip (CLI tests), wave-per-tile A/B, config-5 line, default bench line
mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests/test_gpu_cli.py tests/test_gpu_pipeline.py tests/test_gpu_dropin.py tests/test_gpu_bam.py tests/test_gpu_cli_golden.py -x -q -m gpu > gpurun_out/r3/t_cli.log 2>&1; tail -8 gpurun_out/r3/t_cli.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/wave_tile_ab profiles/tools/wave_tile_ab.hip 2> gpurun_out/r3/wave_build.err && timeout 300 /tmp/wave_tile_ab > gpurun_out/r3/wave_per_tile_ab.txt 2>&1; cat gpurun_out/r3/wave_per_tile_ab.txt
timeout 1200 python bench.py --read-len 250 --corridor 80 --layout se --subs 0.12 --indel-bases 0.03 --sensitive --steps 3 --no-end-to-end --no-cpu-baseline > gpurun_out/r3/bench_config5.log 2> gpurun_out/r3/bench_config5.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_config5.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['sw_gcells_per_s'], j['kernel_ms'])
PY
timeout 1500 python bench.py > gpurun_out/r3/bench_default2.log 2> gpurun_out/r3/bench_default2.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_default2.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['kernel_ms'])
e=j['end_to_end']; print({k:e[k] for k in e if k not in ('cli_log_tail','command')}); print(e['cli_log_tail'])
print(j['cpu_baseline'])
PY
