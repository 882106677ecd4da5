#!/bin/bash
# cumulative kernel time after each pass (pass 2 was fused into pass 1) (KIDMP_DEBUG_STOP profiling aid), for the given workloads
for w in "$@"; do
  line="$w:"
  for s in 1 3 4 5 0; do
    t=$(KIDMP_DEBUG_STOP=$s python bench.py --no-other-workloads --lib kid_amd/libkidmp_prof.so --workload $w --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; print('%.4f'%json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
    line="$line $t"
  done
  echo "$line   (ms after pass 0, pass 1 [blocks D-N], pass 3, pass 4, full)"
done
