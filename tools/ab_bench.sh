#!/bin/bash
# alternate two ASR_DEBUG settings of the bench step on ONE box: bash tools/ab_bench.sh "nt_8ph=0" "" [rounds] [steps]
A="$1"; B="$2"; R=${3:-3}; S=${4:-40}
for i in $(seq 1 $R); do
  for V in "$A" "$B"; do
    ASR_DEBUG="$V" python bench.py --steps $S --warmup 5 --no-census --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ASR_DEBUG=%-24s %.3f ms/step' % ('$V', d['ms_per_step']))"
  done
done
