mkdir -p gpurun_out/r06
{
for i in 1 2; do
echo "== full assembly (BQ_ASM_FUSE=0) run $i"; BQ_ASM_FUSE=0 python tools/c5_time.py all 2>&1 | grep -v failed
echo "== fused (default) run $i"; python tools/c5_time.py all 2>&1 | grep -v failed
done
} > gpurun_out/r06/asm_fuse_time.txt 2>&1
cat gpurun_out/r06/asm_fuse_time.txt
