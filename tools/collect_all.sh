#!/bin/bash
# Developer tool (GPU box, repo root): everything profiles/ is regenerated from -- the rocprofv3 passes of
# tools/collect_profiles.sh plus the bench lines (default, --workload c5, RCCL world size 1, the any-frame-size kernel on every layer,
# the direct kernels) and the 240x320 / latency / SQ-counter files.
#   bash tools/collect_all.sh gpurun_out/prof_rNN rNN ; python tools/summarize_profiles.py gpurun_out/prof_rNN rNN   (the second call, here in the
#   build container, copies the bench lines; on the GPU box profiles/ is part of the scratch copy of the repository)
set -u
R=${1:-gpurun_out/prof}
TAG=${2:-r04}
mkdir -p "$R"
bash tools/collect_profiles.sh "$R" > "$R.log" 2>&1
# the PMC summary first: the bench lines below read roofline.traffic from profiles/${TAG}_pmc_hbm_traffic*.json
python3 tools/summarize_profiles.py "$R" "$TAG" > "$R/summary.log" 2>&1
python3 bench.py > "$R/bench_n1.json" 2> "$R/bench_n1.err"
python3 bench.py --workload c5 > "$R/bench_c5.json" 2> "$R/bench_c5.err"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline > "$R/bench_rccl.json" 2> "$R/bench_rccl.err"
MP_DEBUG=wino43_gen=2 python3 bench.py --no-cpu-baseline > "$R/bench_gen2.json" 2> "$R/bench_gen2.err"
MP_DEBUG=no_winograd python3 bench.py --no-cpu-baseline --steps 10 > "$R/bench_direct.json" 2> "$R/bench_direct.err"
MP_DEBUG=f16_no_res python3 bench.py --workload c5 --no-cpu-baseline > "$R/bench_c5_stream.json" 2> "$R/bench_c5_stream.err"
{ python3 tools/latency.py; MP_DEBUG=splitk_max=1 python3 tools/latency.py | sed -e 's/^/MP_DEBUG=splitk_max=1  /'; python3 tools/bench_layers.py 2 480 640; echo "== bench_layers 2 240 320"; python3 tools/bench_layers.py 2 240 320; } 2>&1 | grep -v amdgpu.ids > "$R/latency.txt"
bash tools/pmc_f16.sh "$R/sq_c5" > "$R/sq_c5.txt" 2>&1
bash tools/pmc_f16.sh "$R/sq_c3" 64 > "$R/sq_c3.txt" 2>&1
# 240x320 (BASELINE configs[0] frame: the deep layers are 30x40, no multiple of the 4x4 tile): B = 64 throughput and single-pair latency,
# with the round-3 routing (conv_wino43.hip + direct kernels for the odd frames; the F(2x2,3x3) kernel it used then is gone) and today's
{ for g in 1 0; do echo "== MP_DEBUG=wino43_gen=$g"; MP_DEBUG=wino43_gen=$g python3 tools/bench_layers.py 64 240 320; MP_DEBUG=wino43_gen=$g python3 tools/latency.py; done; } 2>&1 | grep -v amdgpu.ids > "$R/bench_240x320.txt"
# same-box A/B of this round's fp32 kernel against the previous round's (when its variant library travels with the snapshot:
# multipoint_amd/libmultipoint_hip_exp_r05.so = today's objects with round 5's conv_wino43.o), alternating child processes
if [ -f multipoint_amd/libmultipoint_hip_exp_r05.so ]; then python3 tools/ab_layers.py 5 r05 new 2>&1 | grep -v amdgpu.ids > "$R/ab_layers.txt"; fi
ls -l "$R" | head -40
