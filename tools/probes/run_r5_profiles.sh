# Round-5 profile set at the final code: bench lines (live PMC traffic, per-kernel HBM rates), rocprofv3 kernel-trace stats of the
# same commands, per-kernel HBM traffic tables.  Output under gpurun_out/r5p_*; summaries are copied to profiles/.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
LINES=$O/r5p_bench_lines.jsonl; : > $LINES
run_line() {  # tag, args...
  tag=$1; shift
  python bench.py "$@" 2>$O/r5p_$tag.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); r['tag']='$tag'; print(json.dumps(r))" >> $LINES
  tail -1 $LINES | cut -c1-260
}
run_line c2_b32_b512 --shapes
run_line c2_b32_b512_all_live_captions --caption-tokens 75 --no-cpu-baseline --no-pmc --no-hbm-kernels
MSCLIP_TEXT_PACK=0 run_line c2_b32_b512_full_rows --no-cpu-baseline --no-pmc --no-hbm-kernels
run_line c3_b16_b256 --model b16-yfcc-msclips --batch 256 --no-cpu-baseline
run_line c4rank_b32_b1024 --batch 1024 --no-cpu-baseline --no-pmc
run_line c5_l14_fp8_b256 --model l14-fp8-msclips --batch 256 --no-cpu-baseline --steps 10 --warmup 3
run_line c5_l14_fp8qkv_b256 --model l14-fp8-msclips --precision fp8-qkv --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3
run_line c5_l14_bf16_b256 --model l14-fp8-msclips --precision bf16 --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3
run_line train_b32_b512_bnfrozen --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5
run_line train_b32_b512_bnbatch --train --bn batch --no-cpu-baseline --no-pmc --steps 15 --warmup 5
run_line train_b16_b256_bnbatch --train --bn batch --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 4
cd /tmp; export TMPDIR=/tmp
prof() {  # tag, -- args
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5p_prof_$tag -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc --no-hbm-kernels "$@" > $O/r5p_prof_$tag.log 2>&1
  tail -1 $O/r5p_prof_$tag.log | cut -c1-160
  T=$(find $O/r5p_prof_$tag -name "*kernel_trace.csv" | head -1)
  case $tag in train*) (cd $R; python tools/timeline.py $T --top 40 --gaps 12 > $O/r5p_timeline_$tag.txt 2>&1);; c2_b32_b512) (cd $R; python tools/timeline.py $T --step-marker loss_from_partials_kernel --top 30 --gaps 8 > $O/r5p_timeline_$tag.txt 2>&1);; esac
  find $O/r5p_prof_$tag -name "*kernel_trace.csv" -delete; find $O/r5p_prof_$tag -name "*agent_info.csv" -delete
}
prof c2_b32_b512 --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c2_b32_b512_inline --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c3_b16_b256_inline --model b16-yfcc-msclips --batch 256 --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c5_l14_fp8_b256_inline --model l14-fp8-msclips --batch 256 --steps 10 --warmup 3
prof train_b32_b512 --train --bn batch --steps 8 --warmup 3
prof train_b32_b512_frozen --train --bn frozen --steps 8 --warmup 3
cd $R
MSCLIP_CONV_SIDE_STREAM=0 bash tools/pmc_bench.sh r5c2
MSCLIP_CONV_SIDE_STREAM=0 bash tools/pmc_bench.sh r5c3 --model b16-yfcc-msclips --batch 256
MSCLIP_CONV_SIDE_STREAM=0 bash tools/pmc_bench.sh r5c5 --model l14-fp8-msclips --batch 256
python tools/kernel_rates.py $O/r5p_prof_c2_b32_b512_inline $O/pmcb_r5c2 25 "Per-kernel HBM traffic and rates, C2 (ViT-B/32, batch 512, packed captions), round 5 final code, inline schedule (MSCLIP_CONV_SIDE_STREAM=0)" > $O/r5p_kernel_hbm_rates_b32_b512.md
python tools/kernel_rates.py $O/r5p_prof_c3_b16_b256_inline $O/pmcb_r5c3 25 "Per-kernel HBM traffic and rates, C3 (ViT-B/16, batch 256, packed captions), round 5 final code, inline schedule" > $O/r5p_kernel_hbm_rates_b16_b256.md
python tools/kernel_rates.py $O/r5p_prof_c5_l14_fp8_b256_inline $O/pmcb_r5c5 13 "Per-kernel HBM traffic and rates, C5 (ViT-L/14 fp8, batch 256, packed captions), round 5 final code, inline schedule" > $O/r5p_kernel_hbm_rates_l14_fp8_b256.md
rm -rf $O/pmcb_r5c2_* $O/pmcb_r5c3_* $O/pmcb_r5c5_*
ls $O | grep r5p | head -60
