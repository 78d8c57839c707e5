#!/bin/bash
# kernel stats + one SQ counter pass over tools/conv_first_once.py.  usage (GPU box, repository root): bash tools/profile_conv_first.sh TAG
set -e
TAG=${1:-cf}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/tools/conv_first_once.py > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/sq -o k -- python3 $R/tools/conv_first_once.py > $OUT/sq.log 2>&1
echo "sq pass done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS \
    --output-format csv -d $OUT/sq2 -o k -- python3 $R/tools/conv_first_once.py > $OUT/sq2.log 2>&1 || echo "second counter pass failed"
echo "sq2 pass done"
cd $R
python3 tools/kstat.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) convf > $OUT/${TAG}_kstat.txt
python3 tools/pmc_sq.py $(find $OUT/sq -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_sq.json $OUT/${TAG}_pmc_sq.csv || true
cp $(find $OUT/sq2 -name "*counter_collection.csv" | head -1) $OUT/${TAG}_sq2_raw.csv || true
rm -rf $OUT/stats $OUT/sq $OUT/sq2
cat $OUT/${TAG}_kstat.txt
