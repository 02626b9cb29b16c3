mkdir -p gpurun_out/r06s
export BQHIP_LIBRARY=$PWD/bayesian-quadrature_amd/libbqhip_dbg.so
export REPS=2500
for d in 0 1 4 5 8 16 24; do
  echo "dbg $d"
  BQ_TS_DBG=$d python tools/panel_solve_power.py 2
done > gpurun_out/r06s/ablate.txt 2>&1
cat gpurun_out/r06s/ablate.txt
