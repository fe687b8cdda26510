#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes of their own
# (HBM read / write bytes, SQ MFMA-busy counters).  usage: tools/profile_bench.sh <tag> [bench args]
set -e
tag=$1; shift
ROOT=$(pwd)
out=$ROOT/gpurun_out/prof_bench/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $out/kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $out/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $out/write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/sq -o p -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $out/sq.log 2>&1
cd $ROOT
python3 tools/profile_summary.py $out
