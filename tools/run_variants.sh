#!/bin/bash
# Developer tool (GPU box): per-layer timings of the regular library and of every libmultipoint_hip_exp_*.so named on the
# command line.   tools/run_variants.sh out.log base x1 x2 ...
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"; : > $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset MP_LIB; else export MP_LIB=$PWD/multipoint_amd/libmultipoint_hip_exp_$v.so; fi
  echo "== $v" >> $OUT
  python3 tools/bench_layers.py 64 2>&1 | grep -v "^det\.\|^desc\." >> $OUT
done
