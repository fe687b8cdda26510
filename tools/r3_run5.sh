set -e
out=gpurun_out/r3_t
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread -k "row_col_sums or softmax or xent or no_alloc or argmax" > $out/pytest5.txt 2>&1 || { tail -60 $out/pytest5.txt; exit 1; }
tail -3 $out/pytest5.txt
timeout -k 10 300 python tools/hbm_kernels.py > $out/hbm.jsonl 2>$out/hbm.err || { tail $out/hbm.err; exit 1; }
grep -E "32768|4321" $out/hbm.jsonl | cut -c1-220
