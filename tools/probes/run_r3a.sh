# round 3, first GPU call: tests, then A/B of the new defaults
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3a_tests.log
tail -15 gpurun_out/r3a_tests.log
python bench.py > gpurun_out/r3a_bench_default.json 2> gpurun_out/r3a_bench_default.err; tail -c 2500 gpurun_out/r3a_bench_default.json
for v in "MSCLIP_CONV_SIDE_STREAM=0" "MSCLIP_FULL_LAST_BLOCK=1" "MSCLIP_CONV_SIDE_STREAM=0 MSCLIP_FULL_LAST_BLOCK=1"; do
  echo "== $v"; env $v python bench.py --no-cpu-baseline --no-pmc --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('roofline',{}).get('frac'))"
done
echo "== default again"; python bench.py --no-cpu-baseline --no-pmc --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('roofline',{}).get('frac'), r['roofline'].get('timed_region_overlapped'))"
