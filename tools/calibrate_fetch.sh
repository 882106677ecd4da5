#!/bin/bash
# FETCH_SIZE calibration for this kernel's 8-byte-per-lane loads: with KIDMP_DEBUG_STOP=1 a launch reads
# exactly 10 input profiles (ncol*nz*8 B each) and writes nothing but LDS.
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_cal; KIDMP_DEBUG_STOP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_cal -- python3 bench.py --no-other-workloads --lib kid_amd/libkidmp_prof.so --workload config3 --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2> gpurun_out/pmc_cal.err
python tools/summarise_pmc.py gpurun_out/pmc_cal thompson_column_step gpurun_out/pmc_cal.json | grep FETCH
python -c "
import json; d=json.load(open('gpurun_out/pmc_cal.json')); f=d['mean']['FETCH_SIZE']*1024; true=10*100000*120*8
print('FETCH_SIZE bytes %.4g, true bytes read %.4g, factor true/reported = %.3f'%(f,true,true/f))"
