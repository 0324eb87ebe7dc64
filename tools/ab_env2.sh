#!/bin/bash
# bench iteration time for several environment settings in alternating fresh processes: tools/ab_env2.sh REPS "cfg1" "cfg2" ... ("-" = none)
reps=${1:-3}; shift
for i in $(seq $reps); do
  for cfg in "$@"; do
    c=$cfg; [ "$c" = "-" ] && c=""
    env $c python bench.py --steps 3 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline 2>gpurun_out/ab_env2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${c:-default}:', 'iter ms %.2f' % d['roofline']['ms'], 'step ms %.1f' % d['ms_per_step'], 'deskew ms %.2f' % d['roofline_deskew']['ms'], d.get('alloc_layout'), flush=True)" || tail -3 gpurun_out/ab_env2.err
    grep "bh tune" gpurun_out/ab_env2.err
  done
done
