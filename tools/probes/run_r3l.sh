R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -x -s -k "f8 or fp8 or l16" 2>&1 | grep -v "^$" | tail -12
for rep in 1 2; do echo -n "l16 fp8: "; python bench.py --model l16-fp8-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['achieved'], r['roofline']['launches_per_step'], r['roofline']['time_share_of_step'], r.get('roofline_bf16_gemm',{}).get('time_share_of_step'))"; done
echo -n "b32 fp8: "; python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synth_sd
from msclip_amd import synth
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
for prec in ("bf16", "fp8"):
    m = get_clip_model(named_config("b32-yfcc-msclips", ["MODEL.SPEC.PRECISION", prec])); m.load_state_dict(synth_sd("b32-yfcc-msclips")); m = m.cuda().eval()
    img, tok = synth.synth_images(512, seed=1).cuda(), synth.synth_tokens(512, seed=2).cuda()
    for _ in range(4): m.contrastive_loss(img, tok)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.contrastive_loss(img, tok)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(prec, f"{dt*1e3:.2f} ms/step {512/dt:.0f} pairs/s", end="; ")
PY
