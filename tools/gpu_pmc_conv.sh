#!/bin/bash
# PMC passes over one conv layer micro-benchmark:  bash tools/gpu_pmc_conv.sh <tag> <layer> <pass> <mode>
set -u
TAG=$1; LAYER=$2; PASS=$3; MODE=$4
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
CMD="python tools/conv_bench.py $LAYER $PASS 5 $MODE"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line -f csv -d $O/p$i -- $CMD > $O/p$i.log 2>&1 || echo "pass $i failed: $line"
done <<'EOC'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum
EOC
python tools/pmc_summary.py $O/pmc.json $O/p* > $O/pmc.md 2>&1
rm -rf $O/p[0-9]*/
python - <<EOP
import json
d=json.load(open('$O/pmc.json'))
for k,v in d.items():
    if 'conv' in k:
        print(k)
        for c in sorted(v): print('   %-40s %.4g' % (c, v[c]))
EOP
