#!/bin/bash
# Developer tool (GPU box): s_memtime phase breakdown (tools/conv_timing_wino.py) for -DMP_TIMING variant libraries.
#   tools/run_timing_variants.sh <H selector> t0 t1 ...
H=$1; shift
for v in "$@"; do
  echo "== $v"
  MP_LIB=$PWD/multipoint_amd/libmultipoint_hip_exp_$v.so MP_TIMING_H=$H python3 tools/conv_timing_wino.py 2>&1 | grep "mean\|GHz"
done
