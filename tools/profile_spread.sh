#!/bin/bash
# 100 train steps under rocprofv3 --kernel-trace -> per-step lengths of the eight recurrence launches and the segments between them
# (tools/step_spread.py): does any step run a recurrence in its slow placement-free mode?   bash tools/profile_spread.sh r04
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o k -- python3 $R/bench.py --steps 100 --warmup 5 --no-census --no-cpu-baseline --no-extra > $OUT/trace.log 2>&1
cd $R
python3 tools/step_spread.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) $OUT/${TAG}_step_spread_full.txt
python3 - $OUT/${TAG}_step_spread_full.txt $OUT/${TAG}_step_spread.txt <<'PY'
import sys
rows = [l.split("|") for l in open(sys.argv[1]) if l.startswith("step")]
steps = [float(r[0].split()[1]) for r in rows]
recs = [[float(x) for x in r[1].split()[1:]] for r in rows]
out = open(sys.argv[2], "w")
steps_s = sorted(steps)
out.write("train steps traced: %d   step ms: min %.3f  median %.3f  p90 %.3f  max %.3f\n" % (len(steps), steps_s[0], steps_s[len(steps_s) // 2], steps_s[int(len(steps_s) * 0.9)], steps_s[-1]))
for i in range(8):
    col = sorted(r[i] for r in recs if len(r) == 8)
    out.write("recurrence launch %d (%s): us min %.0f  median %.0f  p90 %.0f  max %.0f\n" % (i, "forward" if i < 4 else "backward", col[0], col[len(col) // 2], col[int(len(col) * 0.9)], col[-1]))
slow = sum(1 for r in recs if len(r) == 8 and (max(r[:4]) > 1400 or max(r[4:]) > 1550))
out.write("steps with a forward launch > 1.40 ms or a backward launch > 1.55 ms: %d of %d\n" % (slow, len(recs)))
PY
rm -rf $OUT/trace
cat $OUT/${TAG}_step_spread.txt
