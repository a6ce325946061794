#!/bin/bash
# build the HIP library (stop on any compile error), then run a command on the GPU box:  tools/dev_gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build(); print('build ok')"
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
