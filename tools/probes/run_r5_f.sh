R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "adapter" 2>&1 | tail -4
python tools/probes/adapter_bench.py 2>&1 | tee $O/r5f_adapter_bench.txt
for i in 1 2; do for f in 0 1; do
MSCLIP_ADAPTER_SAMPLE=$f python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('sample=$f', r['ms_per_step'], r['value'])"
done; done
for f in 0 1; do
MSCLIP_ADAPTER_SAMPLE=$f python bench.py --model b16-yfcc-msclips --batch 256 --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('C3 sample=$f', r['ms_per_step'], r['value'])"
done
