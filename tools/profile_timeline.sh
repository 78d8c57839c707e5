#!/bin/bash
# one rocprofv3 kernel-trace pass of the bench step -> kernel stats + the per-segment timeline (tools/timeline.py): bash tools/profile_timeline.sh r05x
set -e
TAG=${1:-tl}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/bench.py --steps 4 --warmup 2 --no-census --no-cpu-baseline --no-extra > $OUT/stats.log 2>&1
cd $R
python3 tools/prof_summary.py stats $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv > /dev/null
python3 tools/timeline.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/${TAG}_timeline.txt > /dev/null || true
rm -rf $OUT/stats
tail -2 $OUT/stats.log | cut -c1-300
