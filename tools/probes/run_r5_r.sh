R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do for f in 1 0; do
MSCLIP_STREAM_OBJECT=$f python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('stream_object=$f', r['ms_per_step'], r['value'])"
done; done | tee gpurun_out/r5r_stream_ab.txt
