for w in config2 config3 config5; do for c in 1 4; do
 t=$(KIDMP_CPW=$c python bench.py --no-other-workloads --lib kid_amd/libkidmp_prof.so --workload $w --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; print('%.4f'%json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
 echo "$w CPW=$c $t"; done; done
