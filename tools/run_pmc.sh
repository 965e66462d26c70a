#!/bin/bash
# rocprofv3 passes of round 2 (run on the GPU box from the repo root): kernel stats, MFMA-busy / SQ counters, HBM bytes.
# Every PMC pass is its own run with --kernel-trace only (MI355X_MICROARCH.md); the program sits directly after `--`.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/prof_step.py 2"
F="python3 $GRAFT_REPO_ROOT/tools/prof_fps.py 2"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/train/mfma -- $P > $OUT/mfma.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/train/sq -- $P > $OUT/sq.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/train/FETCH_SIZE -- $P > $OUT/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/train/WRITE_SIZE -- $P > $OUT/write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fps/FETCH_SIZE -- $F > $OUT/fps_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/fps/WRITE_SIZE -- $F > $OUT/fps_write.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
echo "pmc rc=$?"
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_counters.py $OUT/train $OUT/pmc_counters.json | tail -20
ls $OUT $OUT/stats | head -30
