#!/bin/bash
# launch floor of the column kernel: KIDMP_DEBUG_STOP=9 returns at once (workgroup dispatch + LDS allocation only)
for c in 4 1; do for s in 9 1; do
 t=$(KIDMP_CPW=$c KIDMP_DEBUG_STOP=$s python bench.py --no-other-workloads --lib kid_amd/libkidmp_prof.so --workload config2 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f %.4f'%(d['roofline']['kernel_ms'], d['ms_per_step']))")
 echo "config2 CPW=$c stop=$s: kernel_ms ms_per_step $t"; done; done
