#!/bin/bash
# kernel-level A/B of pw_gemm experiment variants on ONE box: rocprofv3 kernel stats of 6 train steps per variant
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-abv}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${2:-0 1 2 0 1}; do
  export AMPNET_PW_VARIANT=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$v.$RANDOM -o r -- python3 $GRAFT_REPO_ROOT/scratch/prof_step.py 6 > $O/log_$v.txt 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT
for d in $O/v*; do echo "$d"; grep -E "pw_gemm_kernel<128, 4, 1, true|pw_gemm_kernel<64, 4, 1, false|pw_gemm_kernel<128, 4, 1, false|pw_gemm_kernel<64, 2, 1" $d/r_kernel_stats.csv | awk -F, '{print "   ", $(NF-5), $(NF-4), $1, $2, $3, $4}' | cut -c1-200; done
