#!/bin/bash
# images/s of the headline step at smaller batches and stream counts (is the step bound by what stays in the caches
# between layers?)
for b in 16 32 64; do for s in 1 3; do
  timeout -k 10 120 python bench.py --no-cpu --no-configs --batch $b --streams $s --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b streams $s', d['value'], d['ms_per_step'])" || exit 1
done; done
