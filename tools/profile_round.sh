#!/bin/bash
# rocprofv3 passes behind profiles/: kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs (MI355X_MICROARCH.md).
# usage (on the GPU box, from the repository root): bash tools/profile_round.sh r01_final
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --no-census --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o k -- python3 $R/bench.py --steps 2 --warmup 1 --no-census --no-cpu-baseline --no-extra > $OUT/fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o k -- python3 $R/bench.py --steps 2 --warmup 1 --no-census --no-cpu-baseline --no-extra > $OUT/write.log 2>&1
echo "write pass done"
cd $R
python3 tools/prof_summary.py stats $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python3 tools/timeline.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/${TAG}_timeline.txt || true
python3 tools/prof_summary.py pmc $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_fetch_size.csv
python3 tools/prof_summary.py pmc $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_write_size.csv
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch_size.csv $OUT/${TAG}_pmc_write_size.csv $OUT/${TAG}_pmc_traffic.json
rm -rf $OUT/stats $OUT/fetch $OUT/write
ls -la $OUT
