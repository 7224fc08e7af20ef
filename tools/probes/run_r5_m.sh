R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "table_driven" 2>&1 | tail -15
for i in 1 2; do for f in 0 1; do
MSCLIP_REPACK_TABLE=$f python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('table=$f', r['ms_per_step'], r['value'], r['loss'])"
done; done | tee $O/r5m_repack_ab.txt
