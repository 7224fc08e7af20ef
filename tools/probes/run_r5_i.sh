R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -3
python tools/attn_bench.py 2>&1 | tee $O/r5i_attn_bench.txt
python bench.py --model b16-yfcc-msclips --batch 256 --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('C3', r['ms_per_step'], r['value'])"
python bench.py --model l14-fp8-msclips --batch 256 --steps 10 --warmup 3 --no-pmc --no-cpu-baseline --no-probe --no-hbm-kernels 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('C5 fp8', r['ms_per_step'], r['value'])"
