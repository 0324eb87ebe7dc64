#!/bin/bash
# Where the driver puts the library's gigabyte buffers: R-L iteration time without the audition, the workspace
# allocated by hipMalloc against the virtual-memory API with physical chunks of a chosen size, mapped in order or shuffled.
# Fresh process each.  tools/ab_vmm_alloc.sh REPS "cfg1" "cfg2" ...   (a cfg is a space-separated env assignment list, "-" = none)
reps=${1:-3}; shift
if [ $# -eq 0 ]; then set -- "-" "BH_ALLOC_VMM_MB=2" "BH_ALLOC_VMM_MB=2 BH_ALLOC_VMM_SHUFFLE=1" "BH_ALLOC_VMM_MB=64" "BH_ALLOC_VMM_MB=64 BH_ALLOC_VMM_SHUFFLE=1" "BH_ALLOC_VMM_MB=1024"; fi
for i in $(seq $reps); do
  for cfg in "$@"; do
    c=$cfg; [ "$c" = "-" ] && c=""
    env BH_FC_TUNE_ALLOC=0 $c python bench.py --steps 3 --warmup 1 --no-ops --no-end-to-end --no-cpu-baseline 2>gpurun_out/ab_vmm.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${c:-hipMalloc}:', 'iter ms %.2f' % d['roofline']['ms'], 'step ms %.1f' % d['ms_per_step'], 'deskew ms %.2f' % d['roofline_deskew']['ms'], flush=True)" || tail -3 gpurun_out/ab_vmm.err
  done
done
