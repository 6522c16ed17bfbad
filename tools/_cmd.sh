export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e_parity.py tests/test_gpu_c_abi.py tests/test_gpu_f16.py -q -x -k "nms or keypoint or e2e or accounting or pipeline or abi or topk" 2>&1 | tail -n 3
timeout 600 python tools/fuzz_nms.py 150 33 2>&1 | tail -n 1
for w in c3 c5; do
rm -rf gpurun_out/nmsprof_$w
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nmsprof_$w -o p -- python3 tools/nms_only.py $w > gpurun_out/nmsprof_$w.log 2>&1
done
