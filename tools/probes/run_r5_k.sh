R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
for i in 1 2 3; do for f in 0 1; do
MSCLIP_ATTN_TWO_STREAMS=$f python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('two_streams=$f', r['ms_per_step'], r['value'], r['loss'])"
done; done | tee $O/r5k_attn_two_streams.txt
