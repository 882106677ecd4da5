#!/bin/bash
# Regenerates everything under profiles/ for one round tag (run on the GPU box; results land in gpurun_out/profiles_<tag>):
#   tools/refresh_profiles.sh r04
set -e
export TMPDIR=/tmp
tag=${1:-r04}
out=gpurun_out/profiles_$tag
mkdir -p $out
for w in config3 config5 config2; do
  bash tools/pmc_profile.sh $w ${tag}_$w > /dev/null
  cp gpurun_out/pmc_${tag}_$w.json $out/${tag}_pmc_$w.json
  echo "pmc $w done"
done
# the binary32 code objects on the precision-sweep workload (BASELINE config 5)
for a in p32n f32; do
  bash tools/pmc_profile.sh config5 ${tag}_config5_$a $a > /dev/null
  cp gpurun_out/pmc_${tag}_config5_$a.json $out/${tag}_pmc_config5_$a.json
  echo "pmc config5 $a done"
done
# the bench reads profiles/<tag>_pmc_<workload>[_<arith>].json for the measured HBM traffic / VALU counts (fingerprint-checked)
mkdir -p profiles && cp $out/${tag}_pmc_*.json profiles/
python bench.py > $out/${tag}_bench_default.json
echo "bench default: $(cut -c1-160 $out/${tag}_bench_default.json)"
# rocprofv3 --kernel-trace --stats of the default bench command itself (program directly after `--`), plus the
# average over the K timed launches only (rocprofv3's own average includes the W warm-up launches)
for w in default config3 config5 config2; do
  rm -rf gpurun_out/prof_${tag}_$w
  if [ $w = default ]; then
    # exactly `python bench.py`: three workloads in one process; the headline's 5 warm-up + 20 timed launches come first
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$w -- python3 bench.py > /dev/null 2>&1
    python tools/kernel_stats_timed.py gpurun_out/prof_${tag}_$w 20 "thompson_column_step<2, 120, 4, false, false, false>" 5 > $out/${tag}_${w}_kernel_timed_launches.json
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$w -- python3 bench.py --no-cpu-baseline --no-other-workloads --workload $w > /dev/null 2>&1
    python tools/kernel_stats_timed.py gpurun_out/prof_${tag}_$w 20 > $out/${tag}_${w}_kernel_timed_launches.json
  fi
  cp $(ls -t gpurun_out/prof_${tag}_$w/*/*kernel_stats.csv | head -1) $out/${tag}_${w}_kernel_stats.csv
  head -2 $out/${tag}_${w}_kernel_stats.csv | cut -c1-200
  cat $out/${tag}_${w}_kernel_timed_launches.json | tr -d '\n' | cut -c1-300; echo
done
# per-pass: cumulative times, VALU / SALU instructions and lane utilisation (profiling build of the library)
bash tools/pass_times.sh config3 config5 config2 > $out/${tag}_pass_times.txt
bash tools/lane_util.sh config3 config5 config2 > $out/${tag}_lane_util_per_pass.txt
bash tools/ncol_sweep.sh > $out/${tag}_ncol_sweep.txt
cat $out/${tag}_pass_times.txt $out/${tag}_lane_util_per_pass.txt $out/${tag}_ncol_sweep.txt
