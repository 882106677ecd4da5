#!/bin/bash
# throughput vs batch size on one GPU (tail of partially filled rounds, launch floor)
for w in config2 config3; do for n in 3072 10000 12288 30000 100000 400000; do
 python bench.py --no-other-workloads --workload $w --ncol $n --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w ncol $n: %.3e col-steps/s, kernel %.4f ms, frac %.3f'%(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done; done
