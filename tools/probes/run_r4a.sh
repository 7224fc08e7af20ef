#!/bin/bash
# round 4: per-kernel durations of the one-chain and the two-chain schedule (two_chain_probe.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in full two; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4a_$m -o r4a_$m -- python $R/tools/probes/two_chain_probe.py --mode $m --caps 128 > $R/gpurun_out/r4a_$m.log 2>&1
  tail -2 $R/gpurun_out/r4a_$m.log
done
find $R/gpurun_out -name '*r4a*kernel_stats.csv' | head
