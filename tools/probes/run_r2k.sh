R=$GRAFT_REPO_ROOT; cd $R
for pass in 1 2; do
for v in inline side; do
  if [ "$v" = "inline" ]; then export MSCLIP_CONV_INLINE=1; else export MSCLIP_CONV_INLINE=0; fi
  timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['avg_launch_us'])"
done
done
export MSCLIP_CONV_INLINE=0
timeout 300 python bench.py --no-cpu-baseline --no-pmc --model b16-yfcc-msclips --batch 256 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('b16 side', r['value'], r['ms_per_step'])"
MSCLIP_CONV_INLINE=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --model b16-yfcc-msclips --batch 256 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('b16 inline', r['value'], r['ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
