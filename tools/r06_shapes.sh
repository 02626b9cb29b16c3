mkdir -p gpurun_out/r06s
export REPS=3000
for sh in 1024,448,64 1472,448,64 576,448,64 2048,448,64; do SHAPE=$sh python tools/panel_solve_power.py 2 18; done > gpurun_out/r06s/power5.txt 2>&1
REPS=1000 SHAPE=3712,448,100 python tools/panel_solve_power.py 2 18 >> gpurun_out/r06s/power5.txt 2>&1
cat gpurun_out/r06s/power5.txt
