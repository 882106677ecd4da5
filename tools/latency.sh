#!/bin/bash
# average latencies and in-flight levels of SMEM / VMEM / LDS in the column kernel, plus occupancy
export TMPDIR=/tmp
w=${1:-config3}
for set in "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_LEVEL_WAVES" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_CYCLES" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU"; do
  rm -rf gpurun_out/lat; rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/lat -- python3 bench.py --no-other-workloads --workload $w --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python tools/summarise_pmc.py gpurun_out/lat thompson_column_step | python -c "
import json,sys; d=json.load(sys.stdin)['mean']; print({k: round(v) for k,v in d.items()})"
done
rm -rf gpurun_out/lat
