#!/bin/bash
# Runs on the GPU box (via gpurun): where do the waves of each kernel wait? Two PMC passes over one bench step:
# issue/wait split per wave (SQ_WAIT_ANY = parked on s_waitcnt/barrier, SQ_WAIT_INST_ANY = issue stall, SQ_IFETCH*)
# and the instruction cache (requests / hits / misses). Usage: tools/profile_stalls.sh <tag> [bench args...]
set -u
TAG=${1:-stalls}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-side --no-modes"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU --output-format csv -d $R/$OUT/pmc_wait -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $R/$OUT/bench_pmc_wait.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $R/$OUT/pmc_icache -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $R/$OUT/bench_pmc_icache.log 2>&1
echo done
