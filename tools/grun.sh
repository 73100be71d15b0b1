#!/bin/bash
# build, then hand the command to gpurun (a stale .so on the GPU box has cost a round trip before): tools/grun.sh <timeout> '<command>'
cd "$(dirname "$0")/.." || exit 1
python wgsassign_amd/build.py > /tmp/wgs_build.log 2>&1 || { tail -30 /tmp/wgs_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
