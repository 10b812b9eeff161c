#!/bin/bash
# Round 6: order replay through buckets -- buckets per read (2^13 = round 5, up to 2^15) and the hits per bucket aimed at; stress sub-leg, per-phase times
for cfg in "order_buckets_log2=13,order_bucket_fill=8" "order_buckets_log2=15,order_bucket_fill=8" "order_buckets_log2=15,order_bucket_fill=4" "order_buckets_log2=15,order_bucket_fill=2" "order_buckets_log2=14,order_bucket_fill=4"; do
  echo "== $cfg"
  NGM_HIP_TEST_LIMITS=$cfg NGM_HIP_CS_PHASES=1 timeout 600 python profiles/tools/heavy_leg_only.py --steps 3 --no-cpu-baseline --only repeats > gpurun_out/bf.out 2> gpurun_out/bf.err
  grep "order replay through buckets" gpurun_out/bf.err | tail -2 | cut -c1-420
  python -c "
import json
d=json.loads(open('gpurun_out/bf.out').read().strip().splitlines()[-1])
s=d['half_of_the_reads_from_repeats']; print('%.2fM'%(s['value']/1e6), 'ms/step %.0f'%s['ms_per_step'], 'replay ms', round(s['kernel_ms']['candidate_order_replay_on_its_own_stream']))"
done
