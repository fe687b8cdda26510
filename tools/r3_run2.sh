set -e
out=gpurun_out/r3_t
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_dist.py tests/test_gpu_nnet.py -x -q --timeout 120 --timeout-method thread -k "gradient_group or swbd_shape or forward_tool or sharded" > $out/pytest.txt 2>&1 || { tail -60 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
