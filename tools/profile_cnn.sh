#!/bin/bash
# kernel stats + critical-path timeline of the convolutional configuration (bench.py --config cnn); usage: bash tools/profile_cnn.sh TAG [extra bench args]
set -e
TAG=${1:-r02_cnn}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py --config cnn --steps 4 --warmup 2 --no-census --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
cd $R
python3 tools/prof_summary.py stats $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/stats
