#!/bin/bash
set -o pipefail
O=gpurun_out/r02c
mkdir -p $O
for v in def v2 v2s v2n; do
  lib=$PWD/bayesian-quadrature_amd/libbqhip_$v.so
  [ $v = def ] && lib=$PWD/bayesian-quadrature_amd/libbqhip.so
  BQHIP_LIBRARY=$lib timeout -k 10 120 python tools/potf2_probe.py > $O/potf2_$v.json 2> $O/potf2_$v.err || { echo "probe $v failed"; tail -5 $O/potf2_$v.err; }
  python - <<PY
import json
d=json.load(open('$O/potf2_$v.json'))
k='gauss_from_lds1'
print('$v', round(d[k]['us_per_launch'],2), d[k]['stamp_ticks_load_chain_blocks_tail'], d[k]['errL'])
PY
  BQHIP_LIBRARY=$lib timeout -k 10 120 python tools/c2_timeline.py > $O/timeline_$v.json 2>>$O/potf2_$v.err
  python - <<PY
import json
d=json.load(open('$O/timeline_$v.json'))
print('$v', 'step', d['step_total_mean'], {k:int(v) for k,v in d['phases_mean_steps_1_14'].items()})
PY
  BQHIP_LIBRARY=$lib timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_c2_$v.json 2> $O/bench_c2_$v.err || { echo "bench $v failed"; tail -5 $O/bench_c2_$v.err; }
  python - <<PY
import json
l=json.load(open('$O/bench_c2_$v.json'))
print('$v', 'ms_per_step', l['ms_per_step'], 'parity', l['parity']['logml_rel'], l['parity']['var_rel_prior'])
PY
done
