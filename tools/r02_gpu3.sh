#!/bin/bash
set -o pipefail
O=gpurun_out/r02d
mkdir -p $O
timeout -k 10 120 python tools/potf2_probe.py > $O/potf2.json 2> $O/potf2.err || { echo "probe failed"; tail -5 $O/potf2.err; exit 1; }
python - <<PY
import json, numpy as np
d=json.load(open('$O/potf2.json'))
k='gauss_from_lds1'
print('potf2', round(d[k]['us_per_launch'],2), d[k]['stamp_ticks_load_chain_blocks_tail'], d[k]['errL'], d[k]['err_W'])
a=np.array(d[k]['barrier_arrive_by_wave']); r=np.array(d[k]['barrier_release_by_wave'])
for P in range(0,16,3): print(P, a[P].tolist(), r[P].max())
PY
timeout -k 10 120 python tools/c2_timeline.py > $O/timeline.json 2>>$O/potf2.err
python - <<PY
import json
d=json.load(open('$O/timeline.json'))
print('step', d['step_total_mean'], {k:int(v) for k,v in d['phases_mean_steps_1_14'].items()})
PY
timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err || { echo "bench failed"; tail -5 $O/bench_c2.err; exit 1; }
python - <<PY
import json
l=json.load(open('$O/bench_c2.json'))
print('ms_per_step', l['ms_per_step'], 'parity', l['parity']['logml_rel'], l['parity']['var_rel_prior'])
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
