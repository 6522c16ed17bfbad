export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e_parity.py tests/test_gpu_c_abi.py -q -x -k "nms or keypoint or e2e or accounting or pipeline or abi" 2>&1 | tail -n 4
rm -rf gpurun_out/nmsprof_new
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nmsprof_new -o p -- python3 tools/nms_only.py > gpurun_out/nmsprof_new.log 2>&1
