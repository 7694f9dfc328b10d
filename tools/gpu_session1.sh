#!/bin/bash
# GPU session 1 (baseline of the restored tree): tests, bench, per-layer conv bench, PMC passes.
# Run through gpurun from the repo root; everything lands in gpurun_out/s1/.
set -u
R=$(pwd)
O=$R/gpurun_out/s1
mkdir -p $O
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
timeout 300 python bench.py > $O/bench.log 2>&1
timeout 300 python tools/conv_bench.py > $O/convbench.log 2>&1
# per-kernel trace of the bench
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-conv-events > $O/trace.log 2>&1
# PMC passes (separate runs; --pmc only)
BCMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-events"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $O/pmcA -- $BCMD > $O/pmcA.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -f csv -d $O/pmcB -- $BCMD > $O/pmcB.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $O/pmcC -- $BCMD > $O/pmcC.log 2>&1
python tools/pmc_summary.py $O/pmc.json $O/pmcA $O/pmcB $O/pmcC > $O/pmc.md 2>&1
# keep the merge small: drop the raw per-dispatch CSVs of the PMC passes except a compressed copy
tar czf $O/pmc_raw.tgz -C $O pmcA pmcB pmcC 2>/dev/null
rm -rf $O/pmcA $O/pmcB $O/pmcC
ls -la $O
tail -3 $O/pytest.log; tail -2 $O/bench.log
