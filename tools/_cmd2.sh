for g in 1 0; do
  echo "== MP_WINO43_GEN=$g (1: conv_wino43 + F(2x2) fallback as in round 3; 0: conv_wino43 + conv_wino43b)"
  MP_WINO43_GEN=$g python tools/bench_layers.py 64 240 320 2>&1 | grep -v amdgpu.ids
  MP_WINO43_GEN=$g python tools/latency.py 240 320 2>&1 | grep -v amdgpu.ids
done
