R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "attention or b16 or golden or B16 or oracle" 2>&1 | tail -3
for i in 1 2; do
python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-130
MSCLIP_ATTN_WAVE=1 python bench.py --model b16-yfcc-msclips --batch 256 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-130
done
python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -8
