#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py, and the calibration of
# FETCH_SIZE / WRITE_SIZE for the library's 8-byte-per-lane accesses (tools/calib_stream.py).
# Usage: tools/profile_gpu.sh <tag> [extra workloads...]     e.g.  tools/profile_gpu.sh r04a noise_default lsq_ratio10
# The headline workload (bench.py --workload headline: the timed IOD region only) always gets every pass; each extra
# workload named on the command line (bench.py --workload W: ONLY that region runs in the process, so the counters
# rocprofv3 sums per kernel name belong to it) gets the four passes the bench line's per-workload roofline block needs.
# Never combine --pmc with other trace domains than --kernel-trace (gpurun refuses it), one program after `--`.
set -u
TAG=${1:-r04}; shift || true
EXTRA="$*"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
H="--workload headline"
pass() {  # pass <dir> <bench args> -- <pmc counters...>   (empty counter list = kernel trace + stats)
  local dir=$1 args=$2; shift 2
  if [ $# -eq 0 ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/$dir -- python3 $R/bench.py $args > $R/$OUT/bench_$dir.log 2>&1
  else
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$dir -- python3 $R/bench.py $args > $R/$OUT/bench_$dir.log 2>&1
  fi
  echo "pass $dir done ($(date +%T))"
}
# 1) kernel trace + stats of the bench line's timed region (and, for the record, of the whole default line's kernels)
pass trace "--steps 3 --warmup 1 $H"
# 2) PMC passes of the headline (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one pass)
pass pmc_fetch "--steps 1 --warmup 0 $H" FETCH_SIZE
pass pmc_write "--steps 1 --warmup 0 $H" WRITE_SIZE
pass pmc_sq    "--steps 1 --warmup 0 $H" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS
# 3) fp64 instruction mix
pass pmc_f64   "--steps 1 --warmup 0 $H" SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
# 3a) lane utilisation: active lanes per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU
pass pmc_lanes "--steps 1 --warmup 0 $H" SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES
# 3aa) dynamic instruction classes (round 5): what the VALU instructions that are not fp64 add / mul / fma / trans are,
#      and what issues beside the VALU (scalar ALU, scalar memory, LDS, branches)
pass pmc_mix   "--steps 1 --warmup 0 $H" SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH
# 3b) engine clock under this load: GRBM_GUI_ACTIVE counts GPU clock cycles while the kernel runs (cycles / duration)
pass pmc_clk   "--steps 1 --warmup 0 $H" GRBM_GUI_ACTIVE
# 4) the extra workloads: instruction + lane counters, bytes each way, fp64 mix
for W in $EXTRA; do
  A="--steps 1 --warmup 0 --workload $W"
  pass w_${W}_sq    "$A" SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
  pass w_${W}_fetch "$A" FETCH_SIZE
  pass w_${W}_write "$A" WRITE_SIZE
  pass w_${W}_f64   "$A" SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
done
# 5) calibration of the byte counters for 8-byte-per-lane streams
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/calib_fetch -- python3 $R/tools/calib_stream.py > $R/$OUT/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/calib_write -- python3 $R/tools/calib_stream.py > $R/$OUT/calib_write.log 2>&1
cd $R
python3 -c "import bench; print(bench.source_hash())" > $OUT/source_hash.txt
echo "$EXTRA" > $OUT/extra_workloads.txt
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
