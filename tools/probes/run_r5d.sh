R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -x -k "adapter" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
MSCLIP_CONV_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc --no-hbm-kernels --steps 10 > $O/r5d.log 2>&1
f=$(find /tmp/pa -name "*kernel_stats.csv" | head -1)
grep -i "adapter_gridrow\|ln_stats\|attn_kernel\|front_ws\|dwpool_rows" $f | cut -c1-60,100-200
grep -o '"ms_per_step": [0-9.]*' $O/r5d.log
