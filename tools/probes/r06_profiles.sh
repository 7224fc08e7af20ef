set -x
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
python $R/bench.py --steps 20 --warmup 5 --shapes > $O/c2.json 2> $O/c2.err
python $R/bench.py --steps 20 --warmup 5 --model b16-yfcc-msclips --batch 256 --no-cpu-baseline > $O/c3.json 2> $O/c3.err
python $R/bench.py --steps 20 --warmup 5 --batch 1024 --no-cpu-baseline --no-pmc > $O/c4rank.json 2> $O/c4.err
python $R/bench.py --steps 10 --warmup 3 --model l14-fp8-msclips --batch 256 --no-cpu-baseline > $O/c5.json 2> $O/c5.err
python $R/bench.py --steps 20 --warmup 5 --caption-tokens 75 --no-cpu-baseline --no-pmc > $O/c2_all_live.json 2> $O/c2_all_live.err
MSCLIP_TEXT_PACK=0 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/c2_full_rows.json 2> $O/c2_full_rows.err
MSCLIP_PLAN=0 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/c2_eager.json 2> $O/c2_eager.err
MSCLIP_DYNAMIC_ROWS=0 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/c2_host_rows.json 2> $O/c2_host_rows.err
python $R/bench.py --steps 10 --warmup 3 --train --bn frozen --no-cpu-baseline --no-pmc > $O/train_frozen.json 2> $O/train_frozen.err
python $R/bench.py --steps 10 --warmup 3 --train --bn batch --no-cpu-baseline --no-pmc > $O/train_batch.json 2> $O/train_batch.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-pmc > $O/prof_c2.log 2>&1
MSCLIP_CONV_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2_inline -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-pmc > $O/prof_c2_inline.log 2>&1
find $O -name "*kernel_stats.csv" | head
python $R/tools/timeline.py $(find $O/prof_c2 -name "*kernel_trace.csv" | head -1) --step-marker loss_from_partials_kernel > $O/forward_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -c 300 $O/*.err | tail -40
