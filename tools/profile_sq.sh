#!/bin/bash
# One rocprofv3 --pmc pass of SQ counters (8 SQ slots + GRBM) over a bench.py step -> profiles-ready summaries.
# usage (on the GPU box, from the repository root): bash tools/profile_sq.sh r03
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq -o k -- python3 $R/bench.py --steps 2 --warmup 1 --no-census --no-cpu-baseline --no-extra > $OUT/sq.log 2>&1
echo "sq pass done"
cd $R
python3 tools/pmc_sq.py $(find $OUT/sq -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_sq.json $OUT/${TAG}_pmc_sq.csv
rm -rf $OUT/sq
ls -la $OUT
