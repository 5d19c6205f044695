#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py, and the calibration of
# FETCH_SIZE / WRITE_SIZE for the library's 8-byte-per-lane accesses (tools/calib_stream.py).
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r02}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-side --no-modes"
# 1) kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 $B "$@" > $R/$OUT/bench_trace.log 2>&1
# 2) PMC passes (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $R/$OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $R/$OUT/bench_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/$OUT/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $R/$OUT/bench_pmc_sq.log 2>&1
# 3) fp64 instruction mix, when the counters exist on this device
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $R/$OUT/pmc_f64 -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $R/$OUT/bench_pmc_f64.log 2>&1
# 3a) lane utilisation: active lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $R/$OUT/pmc_lanes -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $R/$OUT/bench_pmc_lanes.log 2>&1
# 3b) engine clock under this load: GRBM_GUI_ACTIVE counts GPU clock cycles while the kernel runs (cycles / duration)
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/pmc_clk -- python3 $R/bench.py --steps 1 --warmup 0 $B "$@" > $R/$OUT/bench_pmc_clk.log 2>&1
# 4) calibration of the byte counters for 8-byte-per-lane streams
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/calib_fetch -- python3 $R/tools/calib_stream.py > $R/$OUT/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/calib_write -- python3 $R/tools/calib_stream.py > $R/$OUT/calib_write.log 2>&1
cd $R
python3 -c "import bench; print(bench.source_hash())" > $OUT/source_hash.txt
find $OUT -name "*.csv" | head -40
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
