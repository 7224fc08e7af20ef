set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_lines_a.jsonl; : > $O
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O
python bench.py --steps 20 --warmup 5 --inline-lengths --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 >> $O
python bench.py --steps 20 --warmup 5 --caption-tokens 75 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 >> $O
MSCLIP_TEXT_PACK=0 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 >> $O
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O
python - <<'PY'
import json
for l in open("gpurun_out/r05_lines_a.jsonl"):
    r=json.loads(l)
    print(r["ms_per_step"], r["value"], r["config"].get("captions","")[:70], "|", r["config"].get("caption_lengths","")[:40], "|", r["config"].get("text_rows","")[:20], "| exec", r["gflop_per_pair"], "frac", r.get("roofline",{}).get("frac"), r.get("roofline",{}).get("isolated",{}).get("frac"), "step", r["whole_step_mfma_frac"], r.get("roofline",{}).get("flops_per_launch_avg"))
PY
