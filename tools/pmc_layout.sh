#!/bin/bash
# Memory-side counters of the bench kernels on the two allocation layouts: TCC <-> DRAM request counts,
# their outstanding level (level / requests = mean latency in TCC clocks) and credit stalls.  Separate passes per counter group.
set -e
for layout in shuffled hipmalloc; do
  if [ $layout = hipmalloc ]; then export BH_ALLOC_VMM_MB=0 BH_VOLUME_POOL=0 BH_FC_TUNE_ALLOC=0; else unset BH_ALLOC_VMM_MB BH_VOLUME_POOL; export BH_FC_TUNE_ALLOC=0; fi
  bash tools/pmc_bench.sh ${layout}_rd "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" --steps 2 --warmup 1 > /dev/null
  bash tools/pmc_bench.sh ${layout}_wr "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" --steps 2 --warmup 1 > /dev/null
  echo "== $layout"; cut -c1-200 gpurun_out/pmc_${layout}_rd/summary.txt | head -8; cut -c1-220 gpurun_out/pmc_${layout}_wr/summary.txt | head -8
done
