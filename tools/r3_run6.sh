set -e
out=gpurun_out/r3_t
mkdir -p $out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/pytest_all.txt 2>&1 || { tail -60 $out/pytest_all.txt; exit 1; }
tail -3 $out/pytest_all.txt
bash tools/profile_hbm.sh r3 > $out/hbm_table.md 2> $out/hbm_table.err || { tail $out/hbm_table.err; exit 1; }
cat $out/hbm_table.md
