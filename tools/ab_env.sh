#!/bin/bash
# A/B of an environment knob on ONE box: rocprofv3 kernel stats of 6 train steps per setting, interleaved.  usage: ab_env.sh OUT VAR "v1 v2" PATTERN
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-abe}; VAR=$2; VALS=${3}; PAT=${4:-bn_finalize}
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in $VALS; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v.$rep -o r -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py 6 > $O/log_$v.txt 2>&1 || { tail -5 $O/log_$v.txt; exit 1; }
done; done
cd $GRAFT_REPO_ROOT
for d in $O/*.[12]; do echo "$d"; python3 tools/step_stats.py $d/r_kernel_stats.csv 0; grep -E "$PAT" $d/r_kernel_stats.csv | awk -F'",' '{print "   ", $2, $1}' | cut -c1-150; done
