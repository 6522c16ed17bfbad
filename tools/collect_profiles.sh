#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are checked against (run on the GPU box from the repo root):
#   1. kernel trace + stats of the bench command            -> gpurun_out/prof/stats_kernel_stats.csv
#   2. PMC FETCH_SIZE and WRITE_SIZE in separate passes     -> gpurun_out/prof/{fetch,write}_counter_collection.csv
#   3. PMC MFMA-busy / wave cycles on the forward pass      -> gpurun_out/prof/mfma_counter_collection.csv
# PMC passes use --kernel-trace only (no sys/hip/hsa trace domains).  tools/summarize_profiles.py turns the CSVs
# into the small files committed under profiles/.
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
BENCH="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
SHORT="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o fetch -- $SHORT > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -o write -- $SHORT > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d "$OUT" -o mfma -- python3 tools/bench_layers.py 64 > "$OUT/mfma.log" 2>&1
ls -l "$OUT"
