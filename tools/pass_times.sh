#!/bin/bash
# cumulative kernel time after each pass (KIDMP_DEBUG_STOP profiling aid), for the given workloads
for w in "$@"; do
  line="$w:"
  for s in 1 2 3 4 5 0; do
    t=$(KIDMP_DEBUG_STOP=$s python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; print('%.4f'%json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
    line="$line $t"
  done
  echo "$line   (ms after pass0..4, full)"
done
