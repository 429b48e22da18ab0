#!/bin/bash
# Developer helper: rebuild the in-tree libraries, then run a script on an MI355X through gpurun.
#   tools/gpu.sh <script.sh> [timeout_s]
set -e
cd "$(dirname "$0")/.."
make -C autoware_vision_pilot_amd/csrc -j8 2>&1 | grep -E "error|Error" && exit 1
make -C adapters > /dev/null
/usr/local/graft/bin/gpurun --timeout "${2:-1500}" -- "bash $1"
