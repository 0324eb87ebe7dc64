#!/bin/bash
# The headline bench in consecutive fresh processes on one box with different allocator layouts: does the slow state of later
# processes (profiles/r04ai_box_state_ab.txt) depend on how the library lays out its gigabyte buffers?
run() { echo -n "$1: "; env $2 timeout -k 10 200 python bench.py --no-ops --no-cpu-baseline --no-end-to-end --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['ms'], 2), 'ms/iteration', d['alloc_layout'].get('chunk_kib'), d['alloc_layout'].get('shuffled'))"; }
run "1 stock" "X=1"
run "2 stock" "X=1"
run "3 hipMalloc (BH_ALLOC_VMM_MB=0)" "BH_ALLOC_VMM_MB=0"
run "4 64-KiB chunks" "BH_ALLOC_VMM_KB=64"
run "5 16-MiB chunks" "BH_ALLOC_VMM_MB=16"
run "6 2-MiB chunks unshuffled" "BH_ALLOC_VMM_SHUFFLE=0"
run "7 stock" "X=1"
