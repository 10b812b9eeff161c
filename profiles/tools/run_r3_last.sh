# round 3: the final tree once more -- smoke(), the whole GPU suite, the default bench line
mkdir -p gpurun_out/r3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3/smoke.log 2>&1; tail -2 gpurun_out/r3/smoke.log
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r3/t_final.log 2>&1; tail -3 gpurun_out/r3/t_final.log
timeout 1500 python bench.py > gpurun_out/r3/bench_final.log 2> gpurun_out/r3/bench_final.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3/bench_final.log') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['kernel_ms'], j['roofline']['frac'], j['roofline']['traffic'])
e=j['end_to_end']; print({k:e[k] for k in e if k not in ('cli_log_tail','command','input')})
PY
