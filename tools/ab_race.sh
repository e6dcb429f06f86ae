#!/bin/bash
# A/B of two builds of the library under tools/race_trace.py on ONE box: is the step body reproducible with each?
# usage: tools/ab_race.sh <control.so> [config batch dtype]     (run from the repo root, through gpurun)
ctl=$1; cfg=${2:-cfg5shard}; b=${3:-4}; dt=${4:-bf16}
for round in 1 2; do
  for lib in "$ctl" ""; do
    echo "== library: ${lib:-shipped}"
    TPGAN_HIP_LIBRARY=$lib REPS=${REPS:-6} timeout -k 10 400 python tools/race_trace.py $cfg $b $dt 2>&1 | grep -v spectral | grep "^run [0-9]*:\|fps call\|cloud " | cut -c1-230 | head -${LINES_MAX:-14} || true
  done
done
