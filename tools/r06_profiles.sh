#!/bin/bash
# Round-6 profile set.  One rocprofv3 run per pass of tools/roofline_run.py (never blended):
# kernel trace + stats for every pass, and -- in separate runs, counters only -- WRITE_SIZE /
# FETCH_SIZE / the MFMA counters for the passes whose rooflines quote them.  The program goes
# directly after `--`.  tools/pmc_summary.py condenses everything into profiles/r06_*.
set -o pipefail
O=$PWD/gpurun_out/r06p
mkdir -p $O
export TMPDIR=/tmp
T="rocprofv3 --kernel-trace --stats --output-format csv"
$T -d $O/bench -o bench -- python3 bench.py --steps 200 --warmup 10 > $O/bench_under_rocprof.json 2> $O/bench.err || { echo bench-prof-failed; tail -5 $O/bench.err; }
echo bench-done
for p in c2 gram potrf256 potrf256_dense potrf_engine trsv solve256 predict c5 c2x256 c3 calib fitpost_n1024 fitpost_n2048 fitpost_n4096 fitpost_n16384; do
  $T -d $O/t_$p -o t -- python3 tools/roofline_run.py $p > $O/t_$p.txt 2>&1 || echo trace-$p-failed
  echo trace-$p-done
done
# the C5 shard once more with the diagonal factors on the main stream (BQ_LOOKAHEAD=0): every
# kernel alone on the chip, its duration what it costs -- in the shipped pass the factors' launches
# sit beside the update and rocprofv3 times their whole residency
BQ_LOOKAHEAD=0 $T -d $O/t_c5_nola -o t -- python3 tools/roofline_run.py c5 > $O/t_c5_nola.txt 2>&1 || echo trace-c5-nola-failed
echo trace-c5-nola-done
for p in gram potrf256 trsv calib c2 solve256; do
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$p -o w -- python3 tools/roofline_run.py $p > $O/w_$p.txt 2>&1 || echo w-$p-failed
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$p -o f -- python3 tools/roofline_run.py $p > $O/f_$p.txt 2>&1 || echo f-$p-failed
  echo pmc-$p-done
done
# MFMA utilisation of the trailing update (gemm_lds_kernel, C4's data and dense operands) and of
# the batched shard
for p in potrf256 potrf256_dense c5; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/m_$p -o m -- python3 tools/roofline_run.py $p > $O/m_$p.txt 2>&1 || echo m-$p-failed
  echo mfma-$p-done
done
python3 tools/pmc_summary.py $O r06 > $O/summary.txt 2>&1 || { echo summary-failed; tail -20 $O/summary.txt; }
# the merge back is capped: keep the small tables only
find $O -name "*agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
python tools/c5_time.py all > $O/c5_time.txt 2>&1 || echo c5-time-failed
for w in c5 c2x256 c3chunk; do python tools/plan_timeline.py $w v > $O/plan_timeline_$w.txt 2>&1 || echo timeline-$w-failed; done
python tools/predict_time.py > $O/predict_time.txt 2>&1 || echo predict-time-failed
python tools/wide_b_check.py > $O/wide_b_check.txt 2>&1 || echo wide-b-failed
python tools/panel_solve_time.py > $O/panel_solve_time.txt 2>&1 || echo panel-solve-time-failed
python tools/trsv_flow_check.py 2048 4096 16384 > $O/trsv_flow_check.txt 2>&1 || echo trsv-flow-failed
python tools/choose_next_time.py > $O/choose_next_time.txt 2>&1 || echo choose-next-failed
python tools/fit_hypers_time.py > $O/fit_hypers_time.txt 2>&1 || echo fit-hypers-failed
# round 6: the panel solve under the socket's power cap (sustained rates, clock, watts), its
# ablations (debug build), and what a hand-off costs inside one XCD
bash tools/r06_shapes.sh > /dev/null 2>&1; cp gpurun_out/r06s/power5.txt $O/panel_solve_power.txt 2>/dev/null || echo shapes-failed
(cd bayesian-quadrature_amd/csrc && make -j8 DEFS=-DBQ_TS_DBG OUT=../libbqhip_dbg.so > /dev/null 2>&1) && bash tools/r06_ablate.sh > /dev/null 2>&1; cp gpurun_out/r06s/ablate.txt $O/panel_solve_ablate.txt 2>/dev/null || echo ablate-failed
rm -f bayesian-quadrature_amd/libbqhip_dbg.so
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O/pmc_sweep_a -o p -- python3 tools/panel_solve_one.py 2 2048 448 64 4 > $O/pmc_sweep_a.txt 2>&1 || echo pmc-sweep-a-failed
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sweep_b -o p -- python3 tools/panel_solve_one.py 2 2048 448 64 4 > $O/pmc_sweep_b.txt 2>&1 || echo pmc-sweep-b-failed
python3 tools/pmc_avg.py trsm_sweep $O/pmc_sweep_a $O/pmc_sweep_b > $O/panel_solve_pmc.txt 2>&1 || echo pmc-sweep-summary-failed
python tools/xcd_hop.py > $O/xcd_hop.txt 2>&1 || echo xcd-hop-failed
python tools/power_probe.py > $O/power_probe.txt 2>&1 || echo power-probe-failed
python bench.py > $O/bench.json 2> $O/bench2.err; echo bench rc=$?
