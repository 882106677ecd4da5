#!/bin/bash
# A/B kernel timing of two builds of libkidmp.so in ONE GPU session (box-to-box noise is ~3 %):
#   tools/ab_compare.sh kid_amd/libkidmp_base.so kid_amd/libkidmp.so [workload ...]
# Alternates the two libraries (bench.py --lib) three times per workload and prints kernel ms per launch.
A=$1; B=$2; shift 2
[ $# -eq 0 ] && set -- config2 config3 config5
for w in "$@"; do
  la=""; lb=""
  for rep in 1 2 3; do
    for which in A B; do
      lib=$A; [ $which = B ] && lib=$B
      t=$(python bench.py --no-other-workloads --lib $lib --workload $w --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; print('%.4f'%json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
      if [ $which = A ]; then la="$la $t"; else lb="$lb $t"; fi
    done
  done
  echo "$w: A=[$la ]  B=[$lb ]"
done
