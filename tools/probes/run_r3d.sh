R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -k "f8 or l16 or gather_rows or last_block or inference_between" -s 2>&1 | tail -25
echo "== glds probe"; timeout 300 python tools/probes/glds_probe.py 2>&1 | tail -14
echo "== bench l16 fp8"; timeout 900 python bench.py --model l16-fp8-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 3 2>&1 | tail -1
echo "== bench l16 bf16"; timeout 900 python - <<'PY' 2>&1 | tail -3
import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
from conftest import synth_sd
from msclip_amd import synth
from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
from msclip_amd.config import named_config
for prec in ("bf16", "fp8"):
    m = get_clip_model(named_config("l16-fp8-msclips", ["MODEL.SPEC.PRECISION", prec])); m.load_state_dict(synth_sd("l16-fp8-msclips")); m = m.cuda().eval()
    img, tok = synth.synth_images(256, seed=1).cuda(), synth.synth_tokens(256, seed=2).cuda()
    for _ in range(3): m.contrastive_loss(img, tok)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.contrastive_loss(img, tok)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(prec, f"{dt*1e3:.2f} ms/step {256/dt:.0f} pairs/s")
    del m
PY
