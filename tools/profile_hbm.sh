#!/bin/bash
# HBM-bound kernels: timing table + FETCH_SIZE / WRITE_SIZE passes (each its own rocprofv3 run).  usage: tools/profile_hbm.sh <tag>
set -e
tag=$1
ROOT=$(pwd)
out=$ROOT/gpurun_out/prof_hbm/$tag
mkdir -p $out
timeout -k 10 200 python3 tools/hbm_kernels.py > $out/timing.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 $ROOT/tools/hbm_kernels.py --pmc > $out/fetch_cases.jsonl 2> $out/fetch.log
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 $ROOT/tools/hbm_kernels.py --pmc > $out/write_cases.jsonl 2> $out/write.log
cd $ROOT
python3 tools/hbm_summary.py $out/timing.jsonl $out/fetch $out/write $out/fetch_cases.jsonl $out/write_cases.jsonl
