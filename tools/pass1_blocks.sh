#!/bin/bash
# Where the column step spends its time, pass by pass and -- inside the rate sweep -- block by block: one experiment build
# per stop (the stop is a compile-time constant, so each truncated kernel is exactly the shipped code up to that point),
# all timed in one process on the workload's initial state (tools/ab_kernel.py --fresh).
#   build (CPU):  tools/pass1_blocks.sh build        -> kid_amd/libkidmp_stop<n>.so, libkidmp_stop0.so = the full step
#   run (GPU):    tools/pass1_blocks.sh run config3 [config5 ...]
# stops: 1 after pass 0; 21 + D snow moments; 22 + E,F,G slopes and warm rain; 23 + H frozen-species rates; 24 + I,J limiters
# and tendencies; 25 + K,L state refresh and snow PSD; 26 + M saturation adjustment; 3 whole sweep (+ N rain evaporation,
# snow fall speed, stores); 4 + pass 3 fall speeds; 5 + pass 4 sedimentation; 0 full step (+ pass 5)
STOPS="1 21 22 23 24 25 26 3 4 5"
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  tools/build_variant.sh stop0 | tail -1
  for s in $STOPS; do tools/build_variant.sh stop$s -DKIDMP_PSTOP_AT=$s | tail -1; done
  exit 0
fi
shift
libs=""
for s in $STOPS 0; do libs="$libs kid_amd/libkidmp_stop$s.so"; done
python tools/ab_kernel.py $libs --workloads "$@" --fresh --steps 5 --reps 3 2>&1 | grep median
