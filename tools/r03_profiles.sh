#!/bin/bash
# Round-3 profile set.  One rocprofv3 run per pass of tools/roofline_run.py (never blended):
# kernel trace + stats for every pass, and -- in separate runs, counters only -- WRITE_SIZE /
# FETCH_SIZE / the MFMA counters for the passes whose rooflines quote them.  The program goes
# directly after `--`.  tools/pmc_summary.py condenses everything into profiles/r03_*.
set -o pipefail
O=$PWD/gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp
T="rocprofv3 --kernel-trace --stats --output-format csv"
$T -d $O/bench -o bench -- python3 bench.py --steps 200 --warmup 10 > $O/bench_under_rocprof.json 2> $O/bench.err || { echo bench-prof-failed; tail -5 $O/bench.err; }
echo bench-done
for p in c2 gram potrf256 potrf_engine trsv solve256 predict c5 c3 calib; do
  $T -d $O/t_$p -o t -- python3 tools/roofline_run.py $p > $O/t_$p.txt 2>&1 || echo trace-$p-failed
  echo trace-$p-done
done
for p in gram potrf256 trsv calib c2 solve256; do
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$p -o w -- python3 tools/roofline_run.py $p > $O/w_$p.txt 2>&1 || echo w-$p-failed
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$p -o f -- python3 tools/roofline_run.py $p > $O/f_$p.txt 2>&1 || echo f-$p-failed
  echo pmc-$p-done
done
# MFMA utilisation of the trailing update (gemm_lds_kernel) and of the batched shard
for p in potrf256 c5; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/m_$p -o m -- python3 tools/roofline_run.py $p > $O/m_$p.txt 2>&1 || echo m-$p-failed
  echo mfma-$p-done
done
python3 tools/pmc_summary.py $O r03 > $O/summary.txt 2>&1 || { echo summary-failed; tail -20 $O/summary.txt; }
# the merge back is capped: keep the small tables only
find $O -name "*agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
BQ_GEMM_TILE=128 python tools/power_probe.py > $O/power_probe_tile128.txt 2>&1 || echo power-128-failed
BQ_GEMM_TILE=64 python tools/power_probe.py > $O/power_probe_tile64.txt 2>&1 || echo power-64-failed
python tools/mfma_sustained.py > $O/mfma_sustained.txt 2>&1 || echo mfma-sustained-failed
python tools/gemm_probe.py > $O/gemm_probe.txt 2>&1 || echo gemm-probe-failed
python bench.py > $O/bench.json 2> $O/bench2.err; echo bench rc=$?
