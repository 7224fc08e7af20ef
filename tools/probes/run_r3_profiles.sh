# Round-3 profile set at the final code: bench lines (with live PMC traffic), rocprofv3 kernel-trace stats of the same commands,
# per-kernel HBM rates, per-shape GEMM traffic.  Output under gpurun_out/r3p_*; the summaries are copied to profiles/ afterwards.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
LINES=$O/r3p_bench_lines.jsonl; : > $LINES
run_line() {  # tag, args...
  tag=$1; shift
  python bench.py "$@" 2>$O/r3p_$tag.err | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); r['tag']='$tag'; print(json.dumps(r))" >> $LINES
  tail -1 $LINES | cut -c1-400
}
run_line c2_b32_b512 --shapes
run_line c3_b16_b256 --model b16-yfcc-msclips --batch 256 --no-cpu-baseline
run_line c4rank_b32_b1024 --batch 1024 --no-cpu-baseline
run_line c5_l16_fp8_b256 --model l16-fp8-msclips --batch 256 --no-cpu-baseline --steps 10 --warmup 3
run_line c5_l16_fp8qkv_b256 --model l16-fp8-msclips --precision fp8-qkv --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3
run_line c5_l16_bf16_b256 --model l16-fp8-msclips --precision bf16 --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3
run_line train_b32_b512_bnbatch --train --bn batch --no-cpu-baseline --steps 15 --warmup 5
run_line train_b32_b512_bnfrozen --train --bn frozen --no-cpu-baseline --no-pmc --steps 15 --warmup 5
run_line train_b16_b256_bnbatch --train --bn batch --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 4
cd /tmp; export TMPDIR=/tmp
prof() {  # tag, env..., -- args
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3p_prof_$tag -o run -- python $R/bench.py --no-cpu-baseline --no-probe --no-pmc "$@" > $O/r3p_prof_$tag.log 2>&1
  tail -1 $O/r3p_prof_$tag.log | cut -c1-160; find $O/r3p_prof_$tag -name "*kernel_trace.csv" -delete; find $O/r3p_prof_$tag -name "*agent_info.csv" -delete
}
prof c2_b32_b512 --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c2_b32_b512_inline --steps 20
prof c3_b16_b256 --model b16-yfcc-msclips --batch 256 --steps 20
MSCLIP_CONV_SIDE_STREAM=0 prof c3_b16_b256_inline --model b16-yfcc-msclips --batch 256 --steps 20
prof c4rank_b32_b1024 --batch 1024 --steps 10
prof c5_l16_fp8_b256 --model l16-fp8-msclips --batch 256 --steps 10 --warmup 3
prof train_b32_b512 --train --bn batch --steps 8 --warmup 3
prof train_b32_b512_frozen --train --bn frozen --steps 8 --warmup 3
# per-kernel HBM traffic of the C2 / C3 steps (inline schedule: counters are per kernel)
cd $R
MSCLIP_CONV_SIDE_STREAM=0 bash tools/pmc_bench.sh r3c2
MSCLIP_CONV_SIDE_STREAM=0 bash tools/pmc_bench.sh r3c3 --model b16-yfcc-msclips --batch 256
bash tools/pmc_bench.sh r3train --train --bn batch
# per-shape traffic of the four projection shapes (isolated launches)
cd /tmp
for shape in qkv out fc proj; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r3p_gemm_${shape}_$c -o run -- python $R/tools/gemm_pmc.py 4 $shape > /dev/null 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
out = {}
for shape in ("qkv", "out", "fc", "proj"):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{O}/r3p_gemm_{shape}_{c}/**/*counter_collection.csv", recursive=True)
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "gemm_pp_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
        v[c] = sum(vals) / len(vals)
    out[shape] = {"fetch_MB_x2": round(2 * v["FETCH_SIZE"] * 1024 / 1e6, 1), "write_MB": round(v["WRITE_SIZE"] * 1024 / 1e6, 1)}
json.dump(out, open(f"{O}/r3p_gemm_shape_traffic.json", "w"), indent=1)
print(out)
PY
ls $O | grep r3p | head -40
