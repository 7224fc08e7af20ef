# training-step timeline: which queue is busy when (tools/timeline.py over a kernel trace of 4 steps)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
for bn in frozen batch; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$bn -o run -- python $R/bench.py --train --bn $bn --steps 4 --warmup 2 --no-cpu-baseline --no-probe --no-pmc > $O/r4j_$bn.log 2>&1
  f=$(find /tmp/tl_$bn -name "*kernel_trace.csv" | head -1)
  python $R/tools/timeline.py $f --gaps 40 > $O/r4j_timeline_$bn.txt 2>&1
  head -12 $O/r4j_timeline_$bn.txt
done
