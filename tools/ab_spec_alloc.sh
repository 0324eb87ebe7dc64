#!/bin/bash
# R-L iteration time (bench.py roofline.ms) in fresh processes: audition of the spectrum allocation (default) against the
# spectrum as half of a double-size allocation (BH_FC_SPEC_X2=1), and against neither (BH_FC_TUNE_ALLOC=0)
for rep in 1 2 3; do
  for v in "" "BH_FC_SPEC_X2=1" "BH_FC_TUNE_ALLOC=0"; do
    env $v python bench.py --steps 3 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${v:-default}', 'iter ms %.2f' % d['roofline']['ms'], 'step ms %.1f' % d['ms_per_step'], 'deskew %.2f fill %.2f' % (d['roofline_deskew']['ms'], d['roofline_deskew']['fill_passes_ms']))"
  done
done
