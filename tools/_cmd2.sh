export MP_WINO43_GEN=2
bash tools/run_variants.sh gpurun_out/r04/gen2_early.txt base early
grep -E "==|conv|total" gpurun_out/r04/gen2_early.txt
