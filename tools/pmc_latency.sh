#!/bin/bash
# usage: tools/pmc_latency.sh <lib.so> <tag> [workload] -- derived latency / occupancy metrics of one library build
export TMPDIR=/tmp
LIB=$1; TAG=$2; W=${3:-config3}
for set in "VmemLatency" "SmemLatency" "LdsLatency" "InstrFetchLatency" "MeanOccupancyPerCU" "MemUnitStalled" "SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-24)
  rm -rf gpurun_out/pl_${TAG}_$n
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pl_${TAG}_$n -- python3 bench.py --no-other-workloads --workload $W --no-cpu-baseline --no-host-entry --steps 5 --warmup 1 --lib $LIB > /dev/null 2> gpurun_out/pl_${TAG}_$n.err
  python tools/summarise_pmc.py gpurun_out/pl_${TAG}_$n thompson_column_step 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$TAG', {k: round(v,2) for k,v in d['mean'].items()})"
done
