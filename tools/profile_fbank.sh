#!/bin/bash
# rocprofv3 kernel stats of the feature kernels alone (tools/time_fbank.py): bash tools/profile_fbank.sh r05_fbank
set -e
TAG=${1:-fbank}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $R/tools/time_fbank.py 20 > $OUT/${TAG}.json 2> $OUT/stats.log
cd $R
python3 tools/prof_summary.py stats $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv > /dev/null
rm -rf $OUT/stats
cat $OUT/${TAG}_kernel_stats.csv
python3 -c "import json,sys; d=json.load(open('$OUT/${TAG}.json')); print({k: d[k] for k in ('us_per_batch','call_wall_us','achieved','frac')})"
