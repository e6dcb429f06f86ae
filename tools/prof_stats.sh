#!/bin/bash
# One rocprofv3 kernel-trace pass of bench.py (cfg2, 10 steps) -> gpurun_out/<dir>/<name>_kernel_stats.csv
#   bash tools/prof_stats.sh <outdir> <name> [bench args...]        (environment variables pass through)
set -eo pipefail
OUT=$1; NAME=$2; shift 2
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out/$OUT"
export TMPDIR=/tmp
d=/tmp/prof_$NAME
rm -rf "$d"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-extra "$@" > "$ROOT/gpurun_out/$OUT/$NAME.log" 2>&1)
cp "$(find "$d" -name '*kernel_stats.csv' | head -1)" "$ROOT/gpurun_out/$OUT/${NAME}_kernel_stats.csv"
rm -rf "$d"
echo "profiled $NAME"
