# every differential check of this directory in one go (on the GPU box); logs under gpurun_out/final_fuzz
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out/final_fuzz
o=gpurun_out/final_fuzz
timeout -k 10 300 python -u tests/fuzz/cumatrix_walk.py --seeds 6 --steps 2500 --out $o/cumatrix_walk.txt > $o/cumatrix_walk.log 2>&1; echo "cumatrix_walk rc $?"; head -1 $o/cumatrix_walk.txt
timeout -k 10 300 python -u tests/fuzz/nnet_walk.py --cases 200 --steps 60 --seed 2 --out $o/nnet_walk.txt > $o/nnet_walk.log 2>&1; echo "nnet_walk rc $?"; head -1 $o/nnet_walk.txt
timeout -k 10 300 python -u tests/fuzz/rbm_walk.py --cases 120 --steps 60 --seed 2 --out $o/rbm_walk.txt > $o/rbm_walk.log 2>&1; echo "rbm_walk rc $?"; head -1 $o/rbm_walk.txt
timeout -k 10 600 python -u tests/fuzz/diff_fuzz.py --cases 150 --seed 11 --siblings --exotic > $o/diff_fuzz.log 2>&1; echo "diff_fuzz rc $?"; tail -2 $o/diff_fuzz.log
timeout -k 10 300 python -u tests/fuzz/op_fuzz.py --cases 600 --seed 5 > $o/op_fuzz.log 2>&1; echo "op_fuzz rc $?"; tail -1 $o/op_fuzz.log
timeout -k 10 300 python -u tests/fuzz/dp_fuzz.py --cases 60 --seed 7 > $o/dp_fuzz.log 2>&1; echo "dp_fuzz rc $?"; tail -1 $o/dp_fuzz.log
timeout -k 10 300 python -u tests/fuzz/shard_fuzz.py --cases 12 --seed 3 > $o/shard_fuzz.log 2>&1; echo "shard_fuzz rc $?"; tail -1 $o/shard_fuzz.log
timeout -k 10 200 python -u tests/fuzz/io_forms_check.py > $o/io_forms.log 2>&1; echo "io_forms rc $?"; tail -1 $o/io_forms.log
timeout -k 10 200 python -u tests/fuzz/error_paths_check.py > $o/error_paths.log 2>&1; echo "error_paths rc $?"; grep -c DIFFERENT $o/error_paths.log
timeout -k 10 200 python -u tests/fuzz/options_check.py > $o/options.log 2>&1; echo "options rc $?"; tail -1 $o/options.log
timeout -k 10 300 python -u tests/fuzz/odd_inputs_check.py > $o/odd.log 2>&1; echo "odd_inputs rc $?"; tail -1 $o/odd.log | cut -c1-120
timeout -k 10 300 python -u tests/fuzz/rbm_tool_stat_check.py > $o/rbm_stat.log 2>&1; echo "rbm_stat rc $?"; tail -1 $o/rbm_stat.log
timeout -k 10 600 python -u tests/fuzz/extreme_shapes_check.py > $o/extreme.log 2>&1; echo "extreme_shapes rc $?"; tail -1 $o/extreme.log
timeout -k 10 300 python -u tests/fuzz/table_mutation.py --cases 1500 --seed 4 > $o/table_mutation.log 2>&1; echo "table_mutation rc $?"; head -1 $o/table_mutation.log
timeout -k 10 300 python -u tests/fuzz/model_mutation.py --cases 600 --seed 4 > $o/model_mutation.log 2>&1; echo "model_mutation rc $?"; head -1 $o/model_mutation.log
timeout -k 10 600 python -u tests/fuzz/archive_mutation.py --cases 200 --seed 4 > $o/archive_mutation.log 2>&1; echo "archive_mutation rc $?"; grep "^archive_mutation" $o/archive_mutation.log
timeout -k 10 600 python -u tests/fuzz/options_mutation.py --cases 400 --seed 4 > $o/options_mutation.log 2>&1; echo "options_mutation rc $?"; grep "^options_mutation" $o/options_mutation.log
timeout -k 10 600 python -u tests/fuzz/specifier_mutation.py --cases 400 --seed 4 > $o/specifier_mutation.log 2>&1; echo "specifier_mutation rc $?"; grep "^specifier_mutation" $o/specifier_mutation.log
