#!/bin/bash
# build, check the ABI on the CPU, then run a command on the MI355X box:  tools/gpu.sh [--timeout S] '<command>'
set -e
cd "$(dirname "$0")/.."
make -C chainer-speech-recognition_amd -j8 2>&1 | grep -iE "error|warning" || true
python -m pytest tests/test_abi.py -x -q 2>&1 | tail -1
TO=900
if [ "$1" == "--timeout" ]; then TO=$2; shift 2; fi
exec /usr/local/graft/bin/gpurun --timeout $TO -- "$@"
