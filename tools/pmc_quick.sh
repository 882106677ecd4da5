#!/bin/bash
# usage: tools/pmc_quick.sh <lib.so> <tag> [workload]  -- two counter passes (instruction mix / wait split) of one library build;
# prints per-wave figures.  For A/B builds (tools/build_variant.sh); the committed profiles come from tools/pmc_profile.sh.
export TMPDIR=/tmp
LIB=$1; TAG=$2; W=${3:-config3}
run() { rm -rf gpurun_out/pq_${TAG}_$1; rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pq_${TAG}_$1 -- python3 bench.py --no-other-workloads --workload $W --no-cpu-baseline --steps 5 --warmup 1 --lib $LIB > /dev/null 2> gpurun_out/pq_${TAG}_$1.err; python tools/summarise_pmc.py gpurun_out/pq_${TAG}_$1 thompson_column_step gpurun_out/pq_${TAG}_$1.json > /dev/null; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU"
run b "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT"
run c "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU"
python3 - <<PY
import json
o={}
for x in "abc":
    d=json.load(open("gpurun_out/pq_${TAG}_%s.json"%x)); o.update(d["mean"]); meta=d["meta"]
w=o["SQ_WAVES"]
print("${TAG} ${W}: VGPR %s LDS %s scratch %s" % (meta.get("VGPR_Count"), meta.get("LDS_Block_Size"), meta.get("Scratch_Size")))
print("  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM %.0f SMEM %.0f | life %.0f quad-cycles: issuing %.0f (VALU busy %.0f, scalar %.0f, LDS %.0f), waiting %.0f, issue-stalled %.0f"
      % (o["SQ_INSTS_VALU"]/w, o["SQ_INSTS_SALU"]/w, o["SQ_INSTS_LDS"]/w, o["SQ_INSTS_VMEM"]/w, o["SQ_INSTS_SMEM"]/w, o["SQ_WAVE_CYCLES"]/w,
         o["SQ_ACTIVE_INST_ANY"]/w, o["SQ_ACTIVE_INST_VALU"]/w, o["SQ_ACTIVE_INST_SCA"]/w, o["SQ_ACTIVE_INST_LDS"]/w, o["SQ_WAIT_ANY"]/w, o["SQ_WAIT_INST_ANY"]/w))
print("  GRBM_GUI_ACTIVE %.0f cycles per launch; VALU busy fraction of the chip %.3f; lane utilisation %.3f"
      % (o["GRBM_GUI_ACTIVE"], o["SQ_ACTIVE_INST_VALU"]*4/(1024*o["GRBM_GUI_ACTIVE"]), o["SQ_THREAD_CYCLES_VALU"]/(64*o["SQ_ACTIVE_INST_VALU"])))
PY
