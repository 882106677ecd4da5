#!/bin/bash
# Regenerates everything under profiles/ for one round tag (run on the GPU box; results land in gpurun_out/):
#   tools/refresh_profiles.sh r01
set -e
export TMPDIR=/tmp
tag=${1:-r01}
out=gpurun_out/profiles_$tag
mkdir -p $out
for w in config2 config3 config5; do
  bash tools/pmc_profile.sh $w ${tag}_$w > /dev/null
  cp gpurun_out/pmc_${tag}_$w.json $out/${tag}_pmc_$w.json
  echo "pmc $w done"
done
# the bench reads profiles/<tag>_pmc_<workload>.json for the measured HBM traffic
mkdir -p profiles && cp $out/${tag}_pmc_*.json profiles/
for w in config2 config3 config5; do
  python bench.py --no-other-workloads --workload $w > $out/${tag}_bench_$w.json
  echo "bench $w: $(cut -c1-200 $out/${tag}_bench_$w.json)"
done
# rocprofv3 --kernel-trace --stats of the bench command itself, per workload (program directly after `--`)
for w in config2 config3 config5; do
  rm -rf gpurun_out/prof_${tag}_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$w -- python3 bench.py --no-other-workloads --no-cpu-baseline --workload $w > /dev/null 2>&1
  cp $(ls -t gpurun_out/prof_${tag}_$w/*/*kernel_stats.csv | head -1) $out/${tag}_${w}_kernel_stats.csv
  head -3 $out/${tag}_${w}_kernel_stats.csv
done
