#!/bin/bash
# instruction-cache behaviour of the column kernel (the code is ~100 KB; the I-cache is shared by CU pairs)
export TMPDIR=/tmp
w=${1:-config3}
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  rm -rf gpurun_out/ic; rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/ic -- python3 bench.py --no-other-workloads --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python tools/summarise_pmc.py gpurun_out/ic thompson_column_step | python -c "
import json,sys; d=json.load(sys.stdin)['mean']; print({k: round(v) for k,v in d.items()})"
done
rm -rf gpurun_out/ic
