#!/bin/bash
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for envs in "X=1" "GPU_BLIT_ENGINE_TYPE=2" "HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0" "DEBUG_CLR_LIMIT_BLIT_WG=4"; do
  i=$((i+1))
  export $envs
  rocprofv3 --memory-copy-trace --kernel-trace --stats --output-format csv -d $root/gpurun_out/d2h_$i -o t -- python3 $root/tools/d2h_probe.py 2>/dev/null | grep "D2H"
  echo "  copies:"; cut -d, -f1-4 $root/gpurun_out/d2h_$i/t_memory_copy_stats.csv 2>/dev/null | tail -n +2
  echo "  kernels:"; grep -i "copyBuffer" $root/gpurun_out/d2h_$i/t_kernel_stats.csv | cut -d, -f1-4
  for e in $envs; do unset ${e%%=*}; done
done
