#!/bin/bash
# kernel-level A/B of two builds of the library on ONE box: rocprofv3 kernel stats of 6 train steps each, interleaved
# usage: bash tools/ab_lib.sh OUTDIR "pattern1|pattern2"
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-abl}
PAT=${2:-pw_gemm_kernel<128, 4, 1, true}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in base new base new; do
  if [ $v = base ]; then export AMPNET_LIB_PATH=$GRAFT_REPO_ROOT/tools/lib_base.so; else if [ -n "$AMPNET_NEW_LIB" ]; then export AMPNET_LIB_PATH=$AMPNET_NEW_LIB; else unset AMPNET_LIB_PATH; fi; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v.$RANDOM -o r -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 6 ${3:-fp32} > $O/log_$v.txt 2>&1 || { tail -5 $O/log_$v.txt; exit 1; }
done
cd $GRAFT_REPO_ROOT
for d in $O/base.* $O/new.*; do echo "$d"; python3 tools/step_stats.py $d/r_kernel_stats.csv 0; grep -E "$PAT" $d/r_kernel_stats.csv | awk -F'",' '{print "   ", $2, $1}' | cut -c1-170; done
