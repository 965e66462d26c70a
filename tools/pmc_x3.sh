#!/bin/bash
# SQ / LDS counters of the split-operand kernels (tools/prof_step.py <steps> f32x3), summarised per kernel: bash tools/pmc_x3.sh <out> [mode]
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcx3}
MODE=${2:-f32x3}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $GRAFT_REPO_ROOT/tools/prof_step.py 2 $MODE"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/t/a -- $P > $O/a.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --output-format csv -d $O/t/b -- $P > $O/b.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS --output-format csv -d $O/t/c -- $P > $O/c.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VMEM_WR --output-format csv -d $O/t/d -- $P > $O/d.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum --output-format csv -d $O/t/e -- $P > $O/e.log 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 profiles/pmc_counters.py $O/t $O/pmc.json | tail -14
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
python3 - <<PY
import json
d=json.load(open("$O/pmc.json"))
for k,v in d.items():
    if "Lb1ELi" in k or "x3" in k or ("pw_gemm" in k and v.get("duration_us",0)>300):
        print(k[:140]); print({kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items()})
PY
