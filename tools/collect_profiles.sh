#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are checked against (run on the GPU box from the repo root):
#   1. kernel trace + stats of the bench command                      -> $OUT/stats_kernel_stats.csv        (+ c5_stats_*)
#   2. PMC FETCH_SIZE and WRITE_SIZE in separate passes               -> $OUT/{fetch,write}_counter_collection.csv (+ c5_*)
#   3. PMC MFMA-busy / wave cycles on the forward pass                -> $OUT/mfma_counter_collection.csv   (+ c5_mfma_*)
# PMC passes use --kernel-trace only (no sys/hip/hsa trace domains).  tools/summarize_profiles.py turns the CSVs
# into the small files committed under profiles/.  The second half repeats everything for `--workload c5`
# (BASELINE configs[4] per-GPU share, fp16 MFMA path).
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
for WL in c3 c5; do
  if [ $WL = c3 ]; then P=""; ARG=""; LAY="64"; else P="c5_"; ARG="--workload c5"; LAY="16 1024 1280 f16"; fi
  BENCH="python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline $ARG"      # (45 launches of each kernel: the first, cold one weighs 2 % in the average)
  SHORT="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARG"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${P}stats -- $BENCH > "$OUT/${P}stats.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o ${P}fetch -- $SHORT > "$OUT/${P}fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o ${P}write -- $SHORT > "$OUT/${P}write.log" 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
      -d "$OUT" -o ${P}mfma -- python3 tools/bench_layers.py $LAY > "$OUT/${P}mfma.log" 2>&1
done
ls -l "$OUT"
