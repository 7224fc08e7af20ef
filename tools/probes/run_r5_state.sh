# Round-5 state check: GPU suite, C2 + training bench lines, rocprofv3 kernel stats of the same commands.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/r5s_pytest.txt; cat $O/r5s_pytest.txt
LINES=$O/r5s_bench_lines.jsonl; : > $LINES
run_line() {  # tag, args...
  tag=$1; shift
  python bench.py "$@" 2>$O/r5s_$tag.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); r['tag']='$tag'; print(json.dumps(r))" >> $LINES
  tail -1 $LINES | cut -c1-400
}
run_line c2_b32_b512
run_line train_b32_b512_bnfrozen --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5
run_line train_b32_b512_bnbatch --train --bn batch --no-cpu-baseline --no-pmc --steps 15 --warmup 5
cd /tmp; export TMPDIR=/tmp
prof() {  # tag, -- args
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5s_prof_$tag -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc "$@" > $O/r5s_prof_$tag.log 2>&1
  tail -1 $O/r5s_prof_$tag.log | cut -c1-160; find $O/r5s_prof_$tag -name "*kernel_trace.csv" -delete; find $O/r5s_prof_$tag -name "*agent_info.csv" -delete
}
prof c2_b32_b512 --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c2_b32_b512_inline --steps 20
prof train_b32_b512_frozen --train --bn frozen --steps 8 --warmup 3
prof train_b32_b512 --train --bn batch --steps 8 --warmup 3
ls $O | grep r5s
