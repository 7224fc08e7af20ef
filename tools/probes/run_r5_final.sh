# Final round-5 check at the final code: GPU suite, C2 + training bench lines, rocprofv3 kernel stats of the same commands, timelines.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/r5z_pytest.txt; cat $O/r5z_pytest.txt
python bench.py --shapes 2>$O/r5z_c2.err | tail -1 > $O/r5z_c2_line.json; cut -c1-200 $O/r5z_c2_line.json
: > $O/r5z_train_lines.jsonl
for bn in frozen batch; do python bench.py --train --bn $bn --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 >> $O/r5z_train_lines.jsonl; done
python -c "
import json
for l in open('$O/r5z_train_lines.jsonl'): r=json.loads(l); print(r['metric'][:70], r['ms_per_step'], r['value'])"
cd /tmp; export TMPDIR=/tmp
prof() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5z_prof_$tag -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc --no-hbm-kernels "$@" > $O/r5z_prof_$tag.log 2>&1
  T=$(find $O/r5z_prof_$tag -name "*kernel_trace.csv" | head -1)
  case $tag in train*) (cd $R; python tools/timeline.py $T --top 40 --gaps 12 > $O/r5z_timeline_$tag.txt 2>&1);; esac
  find $O/r5z_prof_$tag -name "*kernel_trace.csv" -delete; find $O/r5z_prof_$tag -name "*agent_info.csv" -delete; }
prof c2 --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c2_inline --steps 20
prof train_frozen --train --bn frozen --steps 8 --warmup 3
prof train_batch --train --bn batch --steps 8 --warmup 3
head -4 $O/r5z_timeline_train_frozen.txt; head -4 $O/r5z_timeline_train_batch.txt
