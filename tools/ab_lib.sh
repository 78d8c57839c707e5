#!/bin/bash
# alternate two builds of the library under the bench step on ONE box: bash tools/ab_lib.sh exp_libs/<name>/libasr_hip.so [rounds] [steps]
# ("" = the in-tree build).  exp_libs/ is scratch (git-ignored): a variant is built there from a copy of csrc/.
A="$1"; R=${2:-3}; S=${3:-40}
for i in $(seq 1 $R); do
  for V in "$A" ""; do
    if [ -n "$V" ]; then export ASR_HIP_LIB="$PWD/$V"; else unset ASR_HIP_LIB; fi
    python bench.py --steps $S --warmup 5 --no-census --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=%-36s %.3f ms/step' % ('${V:-in-tree}', d['ms_per_step']))"
  done
done
