#!/bin/bash
# usage: tools/prof_gemm.sh <tag> <op> <O> <I> <bm> <bn> <s>   (runs on the GPU box; writes gpurun_out/prof/<tag>)
set -e
tag=$1; shift
ROOT=$(pwd)
out=$ROOT/gpurun_out/prof/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $ROOT/tools/gemm_one.py "$@" 30 > $out/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/p1 -o p1 -- python3 $ROOT/tools/gemm_one.py "$@" 10 > $out/p1.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $out
