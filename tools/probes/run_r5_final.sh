# Final round-5 check at the final code: GPU suite, C2 bench line (+ shapes), rocprofv3 kernel stats of the same command (both schedules).
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/r5z_pytest.txt; cat $O/r5z_pytest.txt
python bench.py --shapes 2>$O/r5z_c2.err | tail -1 > $O/r5z_c2_line.json; cut -c1-300 $O/r5z_c2_line.json
python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 > $O/r5z_train_frozen_line.json; cut -c1-260 $O/r5z_train_frozen_line.json
cd /tmp; export TMPDIR=/tmp
for tag in c2 c2_inline; do
  if [ $tag = c2_inline ]; then export MSCLIP_CONV_SIDE_STREAM=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5z_prof_$tag -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc --no-hbm-kernels --steps 20 > $O/r5z_prof_$tag.log 2>&1
  find $O/r5z_prof_$tag -name "*kernel_trace.csv" -delete; find $O/r5z_prof_$tag -name "*agent_info.csv" -delete
done
ls $O | grep r5z
