#!/bin/bash
# rocprofv3 passes of round 3 (run on the GPU box from the repo root): per-step kernel stats, MFMA-busy / SQ / LDS counters and HBM bytes of
# the fp32 step AND of the bf16_store step, HBM bytes of the FPS legs.  Every PMC pass is its own run with --kernel-trace only
# (MI355X_MICROARCH.md); the program sits directly after `--`.   usage: bash tools/run_pmc.sh [outdir-name] [fp32|bf16|fps|all]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
WHAT=${2:-all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=python3
pmc_set() {    # $1 = tag (directory under $OUT), $2.. = program
  local T=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$T/mfma -- "$@" > $OUT/$T.mfma.log 2>&1 &&
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/$T/sq -- "$@" > $OUT/$T.sq.log 2>&1 &&
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$T/lds -- "$@" > $OUT/$T.lds.log 2>&1 &&
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$T/FETCH_SIZE -- "$@" > $OUT/$T.fetch.log 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$T/WRITE_SIZE -- "$@" > $OUT/$T.write.log 2>&1
}
traffic_only() {
  local T=$1; shift
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$T/FETCH_SIZE -- "$@" > $OUT/$T.fetch.log 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$T/WRITE_SIZE -- "$@" > $OUT/$T.write.log 2>&1
}
rc=0
if [ $WHAT = all ] || [ $WHAT = fp32 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step -o r -- $PY $GRAFT_REPO_ROOT/tools/prof_step.py 20 > $OUT/step.log 2>&1 &&
  pmc_set train $PY $GRAFT_REPO_ROOT/tools/prof_step.py 2 || rc=1
  echo "fp32 rc=$rc"
fi
if [ $WHAT = all ] || [ $WHAT = x3 ]; then
  # round 4: the headline mode (f32x3: three-term bf16 split operands on the MFMA-bound layers)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_f32x3 -o r -- $PY $GRAFT_REPO_ROOT/tools/prof_step.py 20 f32x3 > $OUT/step_f32x3.log 2>&1 &&
  pmc_set train_f32x3 $PY $GRAFT_REPO_ROOT/tools/prof_step.py 2 f32x3 || rc=1
  echo "x3 rc=$rc"
fi
if [ $rc = 0 ] && { [ $WHAT = all ] || [ $WHAT = bf16 ]; }; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_bf16 -o r -- $PY $GRAFT_REPO_ROOT/tools/prof_step.py 20 bf16_store > $OUT/step_bf16.log 2>&1 &&
  pmc_set train_bf16_store $PY $GRAFT_REPO_ROOT/tools/prof_step.py 2 bf16_store || rc=1
  echo "bf16 rc=$rc"
fi
if [ $rc = 0 ] && { [ $WHAT = all ] || [ $WHAT = bf16 ] || [ $WHAT = bf16modes ]; }; then
  # the other two opt-in precision modes: matrix-pipe busy + HBM bytes only (what their roofline objects in the bench line need)
  for m in bf16 bf16_train; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/train_$m/mfma -- $PY $GRAFT_REPO_ROOT/tools/prof_step.py 2 $m > $OUT/train_$m.mfma.log 2>&1 &&
    traffic_only train_$m $PY $GRAFT_REPO_ROOT/tools/prof_step.py 2 $m || rc=1
  done
  echo "bf16 modes rc=$rc"
fi
if [ $rc = 0 ] && { [ $WHAT = all ] || [ $WHAT = fps ]; }; then
  traffic_only fps $PY $GRAFT_REPO_ROOT/tools/prof_fps.py 2 c5 &&
  for c in stage2_256x8192_to_4096 stage1_16x16384_to_8192 stage1_256x16384_to_8192 stream_8x32768_to_8192; do
    traffic_only fps_$c $PY $GRAFT_REPO_ROOT/tools/prof_fps.py 2 $c || rc=1
  done
  echo "fps rc=$rc"
fi
cd $GRAFT_REPO_ROOT
[ -d $OUT/train ] && python3 profiles/pmc_counters.py $OUT/train $OUT/pmc_counters.json | tail -18
[ -d $OUT/train_bf16_store ] && python3 profiles/pmc_counters.py $OUT/train_bf16_store $OUT/pmc_counters_bf16_store.json | tail -12
[ -d $OUT/train_f32x3 ] && python3 profiles/pmc_counters.py $OUT/train_f32x3 $OUT/pmc_counters_f32x3.json | tail -14
[ -d $OUT/train_f32x3 ] && python3 profiles/pmc_traffic.py $OUT/train_f32x3 $OUT/pmc_traffic_f32x3.json
[ -f $OUT/step_f32x3/r_kernel_stats.csv ] && python3 tools/step_stats.py $OUT/step_f32x3/r_kernel_stats.csv 14
FPSARGS=""
[ -d $OUT/fps ] && FPSARGS="--fps $OUT/fps"
for c in stage2_256x8192_to_4096 stage1_16x16384_to_8192 stage1_256x16384_to_8192 stream_8x32768_to_8192; do [ -d $OUT/fps_$c ] && FPSARGS="$FPSARGS --fps-case $c=$OUT/fps_$c"; done
[ -d $OUT/train ] && python3 profiles/pmc_traffic.py $OUT/train $OUT/pmc_traffic.json $FPSARGS
[ -d $OUT/train_bf16_store ] && python3 profiles/pmc_traffic.py $OUT/train_bf16_store $OUT/pmc_traffic_bf16_store.json
for m in bf16 bf16_train; do
  [ -d $OUT/train_$m/FETCH_SIZE ] && python3 profiles/pmc_traffic.py $OUT/train_$m $OUT/pmc_traffic_$m.json
  [ -d $OUT/train_$m/mfma ] && python3 profiles/pmc_counters.py $OUT/train_$m $OUT/pmc_counters_$m.json | tail -4
done
[ -f $OUT/step/r_kernel_stats.csv ] && python3 tools/step_stats.py $OUT/step/r_kernel_stats.csv 12
# the raw per-dispatch CSVs are large: keep the summaries
find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
echo "pmc rc=$rc"
