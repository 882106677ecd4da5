#!/bin/bash
# Per-pass issue/stall profile of the column kernel (profiling build, KIDMP_DEBUG_STOP): cumulative after each pass,
# per wave, in quad-cycles: wave life, VALU / scalar busy, waiting at s_waitcnt, instruction counts.
export TMPDIR=/tmp
for w in "$@"; do
 for s in 1 3 4 5 0; do
  rm -rf gpurun_out/ps; KIDMP_DEBUG_STOP=$s rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/ps -- python3 bench.py --no-other-workloads --lib kid_amd/libkidmp_prof.so --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python tools/summarise_pmc.py gpurun_out/ps thompson_column_step | python -c "
import json,sys; d=json.load(sys.stdin)['mean']; w=d['SQ_WAVES']
print('$w stop $s: life %.0f valu_busy %.0f sca_busy %.0f wait_any %.0f wait_inst %.0f  VALU %.0f SALU %.0f'%(d['SQ_WAVE_CYCLES']/w, d['SQ_ACTIVE_INST_VALU']/w, d['SQ_ACTIVE_INST_SCA']/w, d['SQ_WAIT_ANY']/w, d['SQ_WAIT_INST_ANY']/w, d['SQ_INSTS_VALU']/w, d['SQ_INSTS_SALU']/w))"
 done
done
rm -rf gpurun_out/ps
