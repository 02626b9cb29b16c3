#!/bin/bash
# round-2 GPU session 1: potf2 variants (probe + C2 bench), rsq accuracy, full GPU suite
set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
for v in v0 v1 def; do
  lib=bayesian-quadrature_amd/libbqhip_$v.so
  [ $v = def ] && lib=bayesian-quadrature_amd/libbqhip.so
  echo "== potf2 probe $v" 
  BQHIP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/potf2_probe.py > $O/potf2_$v.json 2> $O/potf2_$v.err || { echo "probe $v failed"; tail -5 $O/potf2_$v.err; exit 1; }
  grep -h "us_per_launch\|us_in_kernel\|stamp" $O/potf2_$v.json | head -12
done
timeout -k 10 120 python tools/probe_rsq.py > $O/rsq.txt 2>&1; cat $O/rsq.txt
for v in v0 v1 def; do
  lib=bayesian-quadrature_amd/libbqhip_$v.so
  [ $v = def ] && lib=bayesian-quadrature_amd/libbqhip.so
  echo "== bench C2 $v"
  BQHIP_LIBRARY=$PWD/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_c2_$v.json 2> $O/bench_c2_$v.err || { echo "bench $v failed"; tail -5 $O/bench_c2_$v.err; exit 1; }
  python -c "
import json,sys
l=json.load(open('$O/bench_c2_$v.json'))
print('$v', 'ms_per_step', l['ms_per_step'], 'parity', l['parity'], 'classes', l['roofline']['class_ms_per_step'])"
done
echo "== pytest gpu"
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; tail -15 $O/pytest.txt; exit $rc
