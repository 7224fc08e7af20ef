R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r3f_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3f_tests.log
tail -8 gpurun_out/r3f_tests.log; grep -E "^FAILED|^ERROR" gpurun_out/r3f_tests.log | head -20
for rep in 1 2; do
for v in "MSCLIP_CONV_SIDE_STREAM=1" "MSCLIP_CONV_SIDE_STREAM=0"; do
  echo "== $v"; env $v python bench.py --no-cpu-baseline --no-pmc --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('roofline',{}).get('frac'))"
done; done
echo "== b16"; python bench.py --no-cpu-baseline --no-pmc --steps 20 --model b16-yfcc-msclips --batch 256 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('roofline',{}).get('frac'))"
echo "== b32 b1024"; python bench.py --no-cpu-baseline --no-pmc --steps 20 --batch 1024 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r.get('roofline',{}).get('frac'))"
