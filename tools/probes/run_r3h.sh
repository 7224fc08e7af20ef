R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -q -x -k "train" > gpurun_out/r3h_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3h_tests.log; tail -6 gpurun_out/r3h_tests.log
for rep in 1 2; do for bn in batch frozen; do
  echo -n "bn=$bn: "; python bench.py --train --bn $bn --no-cpu-baseline --no-pmc --steps 15 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
done; done
echo -n "b16 batch: "; python bench.py --train --bn batch --model b16-yfcc-msclips --batch 256 --no-cpu-baseline --no-pmc --steps 10 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'])"
for rep in 1 2; do echo -n "fwd C2: "; python bench.py --no-cpu-baseline --no-pmc --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['frac'])"; done
