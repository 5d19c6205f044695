#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# 1) kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/$OUT/bench_trace.log 2>&1
# 2) PMC passes (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/$OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/$OUT/bench_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/$OUT/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/$OUT/bench_pmc_sq.log 2>&1
cd $R
find $OUT -name "*.csv" | head -40
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
