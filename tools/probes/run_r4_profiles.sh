# Round-4 profile set at the final code: bench lines (live PMC traffic, per-kernel HBM rates), rocprofv3 kernel-trace stats of the
# same commands, per-kernel HBM traffic, per-shape GEMM traffic.  Output under gpurun_out/r4p_*; summaries are copied to profiles/.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
LINES=$O/r4p_bench_lines.jsonl; : > $LINES
run_line() {  # tag, args...
  tag=$1; shift
  python bench.py "$@" 2>$O/r4p_$tag.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); r['tag']='$tag'; print(json.dumps(r))" >> $LINES
  tail -1 $LINES | cut -c1-300
}
run_line c2_b32_b512 --shapes
MSCLIP_LN_FOLD=0 run_line c2_b32_b512_nofold --no-cpu-baseline --no-pmc --no-hbm-kernels
run_line c3_b16_b256 --model b16-yfcc-msclips --batch 256 --no-cpu-baseline
run_line c4rank_b32_b1024 --batch 1024 --no-cpu-baseline
run_line c5_l14_fp8_b256 --model l14-fp8-msclips --batch 256 --no-cpu-baseline --steps 10 --warmup 3
run_line c5_l14_fp8qkv_b256 --model l14-fp8-msclips --precision fp8-qkv --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3
run_line c5_l14_bf16_b256 --model l14-fp8-msclips --precision bf16 --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3
run_line c5b_l16_fp8_b256 --model l16-fp8-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3
run_line train_b32_b512_bnbatch --train --bn batch --no-cpu-baseline --steps 15 --warmup 5
run_line train_b32_b512_bnfrozen --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5
run_line train_b16_b256_bnbatch --train --bn batch --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 4
cd /tmp; export TMPDIR=/tmp
prof() {  # tag, -- args
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4p_prof_$tag -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc "$@" > $O/r4p_prof_$tag.log 2>&1
  tail -1 $O/r4p_prof_$tag.log | cut -c1-160; find $O/r4p_prof_$tag -name "*kernel_trace.csv" -delete; find $O/r4p_prof_$tag -name "*agent_info.csv" -delete
}
prof c2_b32_b512 --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c2_b32_b512_inline --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c3_b16_b256_inline --model b16-yfcc-msclips --batch 256 --steps 20
prof c4rank_b32_b1024 --batch 1024 --steps 10
MSCLIP_CONV_SIDE_STREAM=0 prof c5_l14_fp8_b256_inline --model l14-fp8-msclips --batch 256 --steps 10 --warmup 3
prof train_b32_b512 --train --bn batch --steps 8 --warmup 3
prof train_b32_b512_frozen --train --bn frozen --steps 8 --warmup 3
cd $R
MSCLIP_CONV_SIDE_STREAM=0 bash tools/pmc_bench.sh r4c2
MSCLIP_CONV_SIDE_STREAM=0 bash tools/pmc_bench.sh r4c5 --model l14-fp8-msclips --batch 256
ls $O | grep r4p | head -40
