R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "colsum or column_sums" 2>&1 | tail -5
python tools/probes/aten_sites.py frozen > $O/r5d_aten_sites.txt 2>&1; tail -50 $O/r5d_aten_sites.txt
for i in 1 2; do
for two in 1 0; do
MSCLIP_COLSUM_TWO_STAGE=$two python bench.py --train --bn frozen --no-cpu-baseline --no-pmc --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('two_stage=$two', r['ms_per_step'], r['value'])"
done; done
