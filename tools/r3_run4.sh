set -e
out=gpurun_out/r3_t
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_gpu_nnet.py -x -q --timeout 120 --timeout-method thread -k "relu or dropout or scheduler" > $out/pytest4.txt 2>&1 || { tail -60 $out/pytest4.txt; exit 1; }
tail -3 $out/pytest4.txt
