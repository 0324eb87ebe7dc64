#!/bin/bash
# R-L iteration time of the stock library against a variant (tools/build_variant.py NAME ...) in alternating fresh processes:
# tools/ab_lib.sh NAME [reps] [extra env, e.g. BH_FC_TUNE_ALLOC=0]
name=$1; reps=${2:-3}; extra=$3
for i in $(seq $reps); do
  for lib in "" "$PWD/biahub_amd/build/variants/libbhcore_$name.so"; do
    env $extra ${lib:+BHCORE_LIB=$lib} python bench.py --steps 3 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${lib:+$name}' or 'stock', '$extra', 'iter ms %.2f' % d['roofline']['ms'], 'step ms %.1f' % d['ms_per_step'])"
  done
done
