#!/bin/bash
# A/B of library builds under tools/fps_wave_trace.py (replayed cfg5shard step) on ONE box.  usage: tools/ab_fps.sh lib1.so lib2.so ...  ("" = shipped)
for lib in "$@" ""; do
  echo "== library: ${lib:-shipped}"
  TPGAN_HIP_LIBRARY=$lib REPS=${REPS:-8} timeout -k 10 400 python tools/fps_wave_trace.py replay 2>&1 | grep "^run" | cut -c1-150
done
