#!/bin/bash
# usage: tools/pmc_profile.sh <workload> <tag> [p64|p32n|f32]   -> gpurun_out/pmc_<tag>_{a,b,c,d,e}.json, gpurun_out/pmc_<tag>.json
export TMPDIR=/tmp
A=${3:-p64}
W="--workload $1 --no-cpu-baseline --steps 5 --warmup 1 --arith $A"
run() { rm -rf gpurun_out/pmc_$2_$1; rocprofv3 --pmc $3 --kernel-trace --output-format csv -d gpurun_out/pmc_$2_$1 -- python3 bench.py --no-other-workloads $W > /dev/null 2> gpurun_out/pmc_$2_$1.err; python tools/summarise_pmc.py gpurun_out/pmc_$2_$1 thompson_column_step gpurun_out/pmc_$2_$1.json > /dev/null; }
run a $2 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU"
run b $2 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT"
run c $2 "FETCH_SIZE"
run d $2 "WRITE_SIZE GRBM_GUI_ACTIVE"
run e $2 "SQ_THREAD_CYCLES_VALU"
python - <<PY
import json
out={}
for x in "abcde":
    d=json.load(open("gpurun_out/pmc_$2_%s.json"%x)); out.update(d["mean"]); out["meta"]=d["meta"] or out.get("meta"); out["dispatches"]=d["dispatches"]
out["workload"]="$1"; out["ncol"]=int(out["meta"]["Grid_Size"])//64
# the code object these counters were measured on (bench.py refuses the profile for any other build)
import sys; sys.path.insert(0, ".")
from kid_amd import ThompsonMP
m = ThompsonMP(iiwarm="$1" == "config2"); out["fingerprint"] = m.kernel_fingerprint("$A"); out["arith"] = "$A"; m.close()
json.dump(out,open("gpurun_out/pmc_$2.json","w"),indent=1)
print(json.dumps(out))
PY
