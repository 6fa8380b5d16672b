#!/bin/bash
# A/B: runs bench.py once per library given on the command line (name=path ...), on the GPU box.
for kv in "$@"; do
  name=${kv%%=*}; path=${kv#*=}
  echo "== $name"
  RTX_LIB_PATH=$path timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.2f  Msamples/s %.1f  trace_ms %.2f  frac %.4f  rays/sample %.3f  exact/ray %.2f' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['rays_per_sample'], d['roofline']['exact_tests_per_launch'] * 485.0 / d['roofline']['sphere_tests_per_launch']))"
done
