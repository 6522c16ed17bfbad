set -u
R=gpurun_out/prof_r02b
bash tools/collect_profiles.sh $R > $R.log 2>&1
python3 bench.py > $R/bench_n1.json 2> $R/bench_n1.err
python3 bench.py --workload c5 > $R/bench_c5.json 2> $R/bench_c5.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline > $R/bench_rccl.json 2> $R/bench_rccl.err
MP_WINO43=0 python3 bench.py --no-cpu-baseline > $R/bench_f22.json 2> $R/bench_f22.err
ls -l $R | head -40
