# a long run of the random differential checks with seeds of its own (on the GPU box); logs under gpurun_out/big_fuzz
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
o=gpurun_out/big_fuzz; mkdir -p $o
timeout -k 10 280 python -u tests/fuzz/cumatrix_walk.py --first-seed 1000 --seeds 12 --steps 3000 --out $o/cumatrix_walk.txt > $o/cumatrix_walk.log 2>&1; head -1 $o/cumatrix_walk.txt
timeout -k 10 280 python -u tests/fuzz/nnet_walk.py --cases 500 --steps 80 --seed 3 --out $o/nnet_walk.txt > $o/nnet_walk.log 2>&1; head -1 $o/nnet_walk.txt; grep "kink" $o/nnet_walk.txt
timeout -k 10 200 python -u tests/fuzz/rbm_walk.py --cases 300 --steps 80 --seed 3 --out $o/rbm_walk.txt > $o/rbm_walk.log 2>&1; head -1 $o/rbm_walk.txt
timeout -k 10 900 python -u tests/fuzz/diff_fuzz.py --cases 300 --seed 21 --siblings --exotic > $o/diff_fuzz.log 2>&1; echo "diff_fuzz (--cases 300 --seed 21 --siblings --exotic): $(tail -1 $o/diff_fuzz.log)"
timeout -k 10 280 python -u tests/fuzz/op_fuzz.py --cases 1500 --seed 9 > $o/op_fuzz.log 2>&1; echo "op_fuzz (--cases 1500 --seed 9): $(tail -1 $o/op_fuzz.log)"
timeout -k 10 280 python -u tests/fuzz/dp_fuzz.py --cases 120 --seed 13 > $o/dp_fuzz.log 2>&1; echo "dp_fuzz (--cases 120 --seed 13): $(tail -1 $o/dp_fuzz.log)"
