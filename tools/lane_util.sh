#!/bin/bash
# VALU lane utilisation of the column kernel, cumulative after each pass (KIDMP_DEBUG_STOP):
#   SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64), plus VALU/SALU instructions per column
export TMPDIR=/tmp
for w in "$@"; do
 for s in 1 3 4 5 0; do
  rm -rf gpurun_out/lu; KIDMP_DEBUG_STOP=$s rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/lu -- python3 bench.py --no-other-workloads --lib kid_amd/libkidmp_prof.so --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python tools/summarise_pmc.py gpurun_out/lu thompson_column_step | python -c "
import json,sys; d=json.load(sys.stdin)['mean']; w=d['SQ_WAVES']
print('$w stop $s: VALU/col %.0f SALU/col %.0f valu_active_cycles/col %.0f thread_cycles/col %.0f lane_util %.3f wave_cycles/col %.0f'%(d['SQ_INSTS_VALU']/w, d['SQ_INSTS_SALU']/w, d['SQ_ACTIVE_INST_VALU']*4/w, d['SQ_THREAD_CYCLES_VALU']*4/w, d['SQ_THREAD_CYCLES_VALU']/(d['SQ_ACTIVE_INST_VALU']*64.0), d['SQ_WAVE_CYCLES']*4/w))"
 done
done
rm -rf gpurun_out/lu
