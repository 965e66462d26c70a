#!/bin/bash
# extra SQ counters of the train step's kernels (LDS / co-execution / issue classes), two passes; output summarised by profiles/pmc_counters.py
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcx}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/prof_step.py 2 ${2:-fp32}"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/t/a -- $P > $O/a.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --output-format csv -d $O/t/b -- $P > $O/b.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS --output-format csv -d $O/t/c -- $P > $O/c.log 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_counters.py $O/t $O/pmc_extra.json | tail -14
