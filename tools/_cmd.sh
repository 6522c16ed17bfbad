export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e_parity.py tests/test_gpu_c_abi.py -q -x -k "nms or keypoint or e2e or accounting or pipeline or abi" 2>&1 | tail -n 3
timeout 600 python tools/fuzz_nms.py 150 21 2>&1 | tail -n 2
rm -rf gpurun_out/nmsprof_new
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nmsprof_new -o p -- python3 tools/nms_only.py > gpurun_out/nmsprof_new.log 2>&1
python3 tools/post_kernels.py 2>&1 | grep "detect_keypoints"
