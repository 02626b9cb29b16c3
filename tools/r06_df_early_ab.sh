mkdir -p gpurun_out/r06
for i in 1 2; do
  echo "== df_early=1 (default) run $i"; python tools/c5_time.py all
  echo "== df_early=0 run $i"; BQ_DF_EARLY=0 python tools/c5_time.py all
done > gpurun_out/r06/df_early_ab.txt 2>&1
cat gpurun_out/r06/df_early_ab.txt
