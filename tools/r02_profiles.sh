#!/bin/bash
# round-2 profile set: rocprofv3 kernel stats of the bench command and of the C5 / C3 tools,
# kernel stats + PMC traffic passes of tools/roofline_run.py
set -o pipefail
O=$PWD/gpurun_out/r02p
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --steps 200 --warmup 10 > $O/bench_under_rocprof.json 2> $O/bench.err || { echo bench-prof-failed; tail -5 $O/bench.err; }
echo bench-done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c5 -- python3 tools/c5_time.py > $O/c5.txt 2>&1 || echo c5-prof-failed
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o c3 -- python3 tools/c3_time.py > $O/c3.txt 2>&1 || echo c3-prof-failed
echo c5c3-done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rf_trace -o rf -- python3 tools/roofline_run.py > $O/rf.txt 2>&1 || echo rf-trace-failed
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/rf_pmc_write -o w -- python3 tools/roofline_run.py >> $O/rf.txt 2>&1 || echo rf-w-failed
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/rf_pmc_fetch -o f -- python3 tools/roofline_run.py >> $O/rf.txt 2>&1 || echo rf-f-failed
echo pmc-done
# keep only the small summaries for the merge back
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
ls -la $O $O/*/* | head -40
python bench.py > $O/bench.json 2> $O/bench2.err; echo bench rc=$?
