#!/bin/bash
# round 4: rocprofv3 kernel-trace stats of the training step (both BatchNorm modes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for bn in frozen batch; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4f_$bn -o run -- python $R/bench.py --train --bn $bn --no-cpu-baseline --no-probe --no-pmc --steps 8 --warmup 3 > $O/r4f_$bn.log 2>&1
  f=$(find $O/r4f_$bn -name '*kernel_stats.csv' | head -1); cp $f $O/r4f_train_${bn}_kernel_stats.csv
  tail -1 $O/r4f_$bn.log | cut -c1-200
done
