#!/bin/bash
# round 6: the tall-tile panel solve -- bits, rates per RT, and where the waves' cycles go
set -o pipefail
O=$PWD/gpurun_out/r06s
mkdir -p $O
export TMPDIR=/tmp
python tools/panel_solve_rt.py check > $O/check.txt 2>&1 || { echo check-failed; tail -20 $O/check.txt; exit 1; }
cat $O/check.txt
python tools/panel_solve_rt.py time > $O/time.txt 2>&1 || { echo time-failed; tail -20 $O/time.txt; exit 1; }
cat $O/time.txt
for mode in 2 18; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O/pmc_a_$mode -o p -- python3 tools/panel_solve_one.py $mode 2048 448 64 4 > $O/pmc_a_$mode.txt 2>&1 || echo pmc-a-$mode-failed
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_b_$mode -o p -- python3 tools/panel_solve_one.py $mode 2048 448 64 4 > $O/pmc_b_$mode.txt 2>&1 || echo pmc-b-$mode-failed
  python3 tools/pmc_kernel.py $O/pmc_a_$mode trsm_sweep > $O/pmc_$mode.txt
  python3 tools/pmc_kernel.py $O/pmc_b_$mode trsm_sweep >> $O/pmc_$mode.txt
  cat $O/pmc_$mode.txt
done
find $O -name "*agent_info.csv" -delete
du -sh $O
