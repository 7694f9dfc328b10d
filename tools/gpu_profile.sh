#!/bin/bash
# Profile of the bench command for profiles/: kernel trace (+stats) and three PMC passes.
#   bash tools/gpu_profile.sh <tag>      -> gpurun_out/<tag>/{trace.md, pmc.md, pmc.json, bench.log}
set -u
TAG=${1:-prof}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench.log 2>&1
BCMD="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-conv-events --eager --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- $BCMD > $O/trace.log 2>&1
python tools/summarize_trace_csv.py $O/trace 7 > $O/trace.md
python tools/timeline.py $O/trace 1 > $O/timeline.md 2>&1; cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null; rm -rf $O/trace
PCMD="python bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-conv-events --eager --no-extras"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $O/pmcA -- $PCMD > $O/pmcA.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $O/pmcB -- $PCMD > $O/pmcB.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $O/pmcC -- $PCMD > $O/pmcC.log 2>&1
python tools/pmc_summary.py $O/pmc.json $O/pmcA $O/pmcB $O/pmcC > $O/pmc.md 2>&1
rm -rf $O/pmcA $O/pmcB $O/pmcC
tail -1 $O/bench.log | cut -c1-300; head -8 $O/trace.md
