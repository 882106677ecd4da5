#!/bin/bash
# dynamic VALU instructions per column after each pass (KIDMP_DEBUG_STOP), for one workload
export TMPDIR=/tmp
w=$1
for s in 1 3 4 5 0; do
  rm -rf gpurun_out/vpp; KIDMP_DEBUG_STOP=$s rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/vpp -- python3 bench.py --no-other-workloads --lib kid_amd/libkidmp_prof.so --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python tools/summarise_pmc.py gpurun_out/vpp thompson_column_step | python -c "
import json,sys; d=json.load(sys.stdin)['mean']; w=d['SQ_WAVES']
print('$w stop $s: VALU/col %.0f SALU/col %.0f wave_cycles/col %.0f valu_active/col %.0f wait %.2f'%(d['SQ_INSTS_VALU']/w, d['SQ_INSTS_SALU']/w, d['SQ_WAVE_CYCLES']*4/w, d['SQ_ACTIVE_INST_VALU']*4/w, d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES']))"
done
