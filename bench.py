"""Headline benchmark: image-text pairs/s of the MS-CLIP-S ViT-B/32 bf16 forward + contrastive step
(BASELINE.json configs[1]: batch 512 per GPU, synthetic 224x224 images + 77-token captions, random-init weights).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch per GPU with inputs resident in HBM: both towers, projection +
L2 norm, feature all-gather (RCCL, N > 1), local row/column logits blocks, symmetric cross-entropy (+ scalar
all-reduce).  Weak scaling: the per-GPU batch is fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import msclip_amd                                                   # noqa: E402

msclip_amd.configure_runtime()                                      # HSA_KERNARG_POOL_SIZE, before the first HIP call (its docstring says why)
import torch                                                        # noqa: E402
import torch.distributed as dist                                    # noqa: E402

GFLOP_PER_PAIR = {"b32-yfcc-msclips": 23.549, "b16-yfcc-msclips": 49.617,    # SURVEY.md s8(d), counted on the reference
                  "l16-fp8-msclips": 172.436,   # torch.utils.flop_counter on the reference built from experiments/model/l16-fp8-msclips.yaml (image 125.345 + text 47.091)
                  "l14-fp8-msclips": 209.116}   # the same for experiments/model/l14-fp8-msclips.yaml, tools/make_golden.py --l14 (image 162.026 + text 47.091)
WIDTH = {"b32-yfcc-msclips": 768, "b16-yfcc-msclips": 768, "l16-fp8-msclips": 1024, "l14-fp8-msclips": 1024}
# The reference computes out_proj / c_fc / c_proj of the LAST block on every token although only x[:, 0] (M.py:2685) and the
# EOT row (M.py:3057-3060) are read afterwards; the engine runs them on those rows only (engine._last_block_tail).  FLOPs it
# does not execute are not credited: 18 d^2 per skipped row (2 d^2 out_proj + 8 d^2 c_fc + 8 d^2 c_proj), d = 768.
SKIPPED_ROWS_PER_PAIR = {"b32-yfcc-msclips": 50 + 77 - 2, "b16-yfcc-msclips": 197 + 77 - 2, "l16-fp8-msclips": 197 + 77 - 2,
                         "l14-fp8-msclips": 257 + 77 - 2}
PEAK_FP8_TFLOPS = 5000.0                                                       # MI355X_MICROARCH.md: dense fp8 (MX K = 128) MFMA
PEAK_BF16_TFLOPS = 2500.0                                                      # MI355X_MICROARCH.md: dense bf16 MFMA
DOMINANT = {"variant": "pp", "kernel": "gemm_pp_kernel<0, false",              # what hip.gemm_variant calls it / rocprof's name (prefix: <0, false, 0 | 1 | 2> = standard / LayerNorm-fold consumer / producer epilogues of one main loop)
            "what": "dense bf16 MFMA GEMM: all transformer projections + wide pointwise convs"}
N_SIMD, N_XCD = 1024, 8                                                        # 256 CUs x 4 SIMDs; GRBM counters sum over XCDs


def pmc_passes(args, kernel):
    """HBM traffic and MFMA-pipe occupancy of the dominant kernel inside THIS workload, from rocprofv3 counters:
    separate --pmc passes of a short child run of the same command (FETCH_SIZE and WRITE_SIZE cannot share a pass;
    MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only.  Corrections per that guide's HBM section:
    FETCH_SIZE (KiB) x 2 (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B), WRITE_SIZE as
    reported.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs.  Profiled passes run at
    a lower clock: only the counters are taken from them, never a time."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    out = {}
    sets = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"], "mfma": ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]}
    for tag, counters in sets.items():
        d = tempfile.mkdtemp(prefix=f"msclip_pmc_{tag}_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "run", "--",
               sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--batch", str(args.batch),
               "--model", args.model, "--no-cpu-baseline", "--no-probe", "--no-pmc"]
        if args.caption_tokens:
            cmd += ["--caption-tokens", str(args.caption_tokens)]
        if args.inline_lengths:
            cmd += ["--inline-lengths"]
        if args.precision:
            cmd += ["--precision", args.precision]
        if args.train_slice:                        # the counters of a training record come from a training child
            cmd += ["--train", "--bn", args.bn]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", MSCLIP_CONV_SIDE_STREAM="0"), timeout=900,
                           capture_output=True, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            vals, tot = {}, {}
            for r in csv.DictReader(open(files[0])):
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                if short(r["Kernel_Name"]).startswith(kernel):
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for c in counters:
                out[c] = sum(vals[c]) / len(vals[c])
                out["ALL_" + c] = tot[c]                      # summed over every kernel of the child run
            out["launches_" + tag] = len(vals[counters[0]])
        except Exception as e:                      # a box without counter access still gets its bench line
            out["error_" + tag] = f"{type(e).__name__}: {str(e)[:200]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out


HBM_PEAK_TBPS = 8.0                                                           # MI355X_MICROARCH.md: HBM3E spec peak (6.3 measured achievable)


def hbm_kernel_rates(args, B, width, Lv, Lt, g, Mt_live=None, Mt_rows=None):
    """Achieved HBM rate of the largest bandwidth-bound kernels of the step ("per-kernel HBM GB/s for the
    bandwidth-bound kernels"): average launch time from a rocprofv3 --kernel-trace child pass of this command (inline
    schedule: nothing else on the chip), algorithmic bytes per launch from the shapes (s8(d) "algorithmic bytes")."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    Mv, Mt, D = B * Lv, B * Lt, width
    Mt_live = Mt if Mt_live is None else Mt_live      # packed captions: the rows that exist / the rows the GEMM tiles cover
    Mt_rows = Mt if Mt_rows is None else Mt_rows
    txt = f"text attention core (causal, packed captions: {Mt_live / B:.1f} live rows of {Lt} on average): q|k|v read + output written, bf16"
    alg = {   # kernel-name prefix -> (what, algorithmic bytes per launch)
        "attn_kernel<3, true": (txt, 4 * Mt_live * D * 2),
        "attn_kernel<2, true": (txt, 4 * Mt_live * D * 2),
        "attn_kernel<1, true": (txt, 4 * Mt_live * D * 2),
        "attn_kernel<2, false": ("image attention core (50 tokens)", 4 * Mv * D * 2),
        "attn_wg_kernel": ("image attention core (197 tokens)", 4 * Mv * D * 2),
        "ln_pair_kernel": ("LayerNorm pass (row count differs per call site since the LayerNorm fold: not rated)", None),
        "ln_stats_kernel": ("LayerNorm pass of one tower's rows (+ fold state; adapter layers also copy the fp32 row)", None),
        "front_ws_kernel<0": ("stem conv1 + parallel stage 0 + stem stage 0 in one pass: fp32 image in, two bf16 maps out",
                              B * (3 * 224 * 224 * 4 + 112 * 112 * (D // 16) * 2 + 56 * 56 * (D // 8) * 2)),
        "adapter_gridrow_kernel": ("lateral adapter bottom half + sum + ln_adapt + the block's ln_1: x, t fp32 in, fp32 stream + bf16 operand out",
                                   Mv * D * (4 * 3 + 2)),
        "adapter_sample_kernel": ("lateral adapter bottom half + sum + ln_adapt + the block's ln_1 (workgroup per sample): x, t fp32 in, fp32 stream + bf16 operand out",
                                  Mv * D * (4 * 3 + 2)),
    }
    d = tempfile.mkdtemp(prefix="msclip_ktrace_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "run", "--",
           sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "2", "--batch", str(args.batch),
           "--model", args.model, "--no-cpu-baseline", "--no-probe", "--no-pmc", "--no-hbm-kernels"]
    if args.caption_tokens:
        cmd += ["--caption-tokens", str(args.caption_tokens)]
    if args.precision:
        cmd += ["--precision", args.precision]
    try:
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", MSCLIP_CONV_SIDE_STREAM="0"), timeout=900,
                       capture_output=True, check=True)
        f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
        rows = []
        table = [(r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""), r) for r in csv.DictReader(open(f))]
        # passes in the trace = launches of a once-per-pass kernel (3 steps + 2 warm-up; an uncalibrated fp8 model adds its
        # calibration pass): the loss kernel runs once per step, the text head's L2 norm once per tower and pass
        marks = [int(r["Calls"]) for n, r in table if n.startswith("loss_from_partials_kernel")]
        passes = max(marks[0] if marks else 5, 1)
        for name, r in table:
            for pre, (what, nbytes) in alg.items():
                if name.startswith(pre) and nbytes:
                    us = float(r["AverageNs"]) / 1e3
                    per_step = round(int(r["Calls"]) / passes, 2)
                    rows.append({"kernel": name.split("(")[0], "what": what, "launches_per_step": per_step,
                                 "avg_us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 1),
                                 "achieved_TBps": round(nbytes / us / 1e6, 2), "frac_of_hbm_peak": round(nbytes / us / 1e6 / HBM_PEAK_TBPS, 3),
                                 "step_share_us": round(us * per_step, 1)})
        rows.sort(key=lambda r: -r["step_share_us"])
        return rows[:4]
    except Exception as e:
        return [{"error": f"{type(e).__name__}: {str(e)[:200]}"}]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def load_schema(name):
    with open(os.path.join(ROOT, "tests", "golden", name + ".schema.json")) as f:
        return [(k, tuple(s), getattr(torch, d)) for k, s, d in json.load(f)]


def cpu_baseline(name, sd, batch=32, iters=3):
    """The CPU oracle (a restatement of the reference's forward; the reference's Python cannot travel) timed on this
    box's host cores: forward(image, text) + symmetric CE, fp32, on a bounded sample."""
    from msclip_amd import synth
    from oracle import msclip_oracle as O
    from msclip_amd.zeroshot import effective_cores
    cores = min(effective_cores(), 32)       # usable cores (affinity + cgroup quota); more than 32 intra-op threads only adds barrier overhead
    torch.set_num_threads(cores)
    arch = (O.arch_b32() if name.startswith("b32") else O.arch_l16() if name.startswith("l16") else
            O.arch_l14() if name.startswith("l14") else O.arch_b16())
    img, tok = synth.synth_images(batch, seed=3), synth.synth_tokens(batch, seed=4)
    with torch.no_grad():
        O.contrastive_loss(O.forward(img[:2], tok[:2], sd, arch))            # warm-up
        t0 = time.perf_counter()
        for _ in range(iters):
            O.contrastive_loss(O.forward(img, tok, sd, arch))
        dt = (time.perf_counter() - t0) / iters
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.lower().startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": round(batch / dt, 3), "unit": "pairs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "host_cores_visible": os.cpu_count(), "host_cores_usable": effective_cores(),
            "sample": f"{iters} x forward+loss of {batch} pairs, fp32 torch CPU oracle, {torch.get_num_threads()} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (BASELINE config C2: 512)")
    ap.add_argument("--model", default="b32-yfcc-msclips")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--no-hbm-kernels", action="store_true", help="skip the kernel-trace child pass (per-kernel HBM rates of the bandwidth-bound kernels)")
    ap.add_argument("--prefill-random", action="store_true",
                    help="probe runs only: fill the workspace with random data first (timing ablation builds of the "
                         "library whose kernels skip their stores; zero-filled operands would raise the clock)")
    ap.add_argument("--train", "--train-slice", dest="train_slice", action="store_true",
                    help="time forward + backward + AdamW of the training step (msclip_amd.train: every parameter gets a "
                         "gradient; BatchNorm with frozen running statistics) instead of the forward step -- a separate "
                         "metric, never the headline")
    ap.add_argument("--precision", choices=("bf16", "fp8", "fp8-qkv"), default=None,
                    help="override MODEL.SPEC.PRECISION of the config (fp8: c_fc / c_proj on the fp8 MFMA; fp8-qkv: in_proj as well)")
    ap.add_argument("--bn", choices=("batch", "frozen"), default="batch",
                    help="--train only: train-mode BatchNorm with per-GPU batch statistics (default, the reference's train() "
                         "semantics) or frozen running statistics")
    ap.add_argument("--shapes", action="store_true", help="add the per-shape table of the dominant kernel to the record")
    ap.add_argument("--inline-lengths", action="store_true",
                    help="packed captions: let the engine compute each batch's lengths at the start of its step (host waits for the "
                         "read-back) instead of staging them one step ahead")
    ap.add_argument("--caption-tokens", type=int, default=0,
                    help="every caption has exactly this many content tokens (75 = all 77 rows live: the line that shows what the "
                         "step costs when packing removes nothing); default 0 = SURVEY s8(d)'s synthetic captions, U{4..60} content tokens")
    args = ap.parse_args()

    from msclip_amd import comm as C, hip, synth
    from msclip_amd.clip_openai_pe_res_v1 import get_clip_model
    from msclip_amd.config import named_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MSCLIP_TEST_SHARED_GPU=1 (tests only, tests/test_gpu_model.py): all ranks on GPU 0 over gloo, so the N > 1 plumbing of
    # this script (rendezvous, barriers, max-over-ranks timing, rank-0 record) runs on a one-GPU box.  Never a measurement.
    shared = os.environ.get("MSCLIP_TEST_SHARED_GPU", "0") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    C.init_distributed("gloo" if shared else "nccl")
    rank = C.comm.rank
    if world > 1 and not shared:                      # the collectives really are RCCL over all N ranks
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == args.gpus, \
            (dist.get_backend(), dist.get_world_size(), args.gpus)
        assert torch.cuda.device_count() >= args.gpus, f"--gpus {args.gpus} but only {torch.cuda.device_count()} devices are visible"
    if world > 1:                                     # ... and a collective over them counts every rank exactly once, on every rank
        probe_t = torch.ones(1, device=torch.device("cuda", local))
        dist.all_reduce(probe_t)
        assert int(probe_t.item()) == args.gpus, (int(probe_t.item()), args.gpus)
    dev = torch.device("cuda", local)
    # a process group also exists at N = 1 under MSCLIP_COLLECTIVES_AT_WORLD_1=1 (one rank, every collective through RCCL as an
    # identity: what the collectives cost a step before any wire time; tests/test_gpu_model.py)
    grouped = dist.is_initialized()

    sd = synth.synth_state_dict(load_schema(args.model), seed=0)
    model = get_clip_model(named_config(args.model, ["MODEL.SPEC.PRECISION", args.precision] if args.precision else None))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    eng = model.engine()
    B = args.batch
    img = synth.synth_images(B, seed=10 + rank).to(dev)                      # fp32 pixels (reference API), resident in HBM
    if args.caption_tokens:
        tok = synth.synth_tokens(B, seed=100 + rank, min_len=args.caption_tokens, max_len=args.caption_tokens).to(dev)
    else:
        tok = synth.synth_tokens(B, seed=100 + rank).to(dev)
    lens_host = (tok.argmax(dim=-1) + 1).cpu()                               # live rows per caption (EOT position + 1, M.py:3057)

    ts = None
    if args.train_slice:
        from msclip_amd import train
        ts = train.from_config(model, named_config(args.model), bn=args.bn)

    # Caption lengths.  With packed text rows the engine needs one host-side number per batch (the total of live rows).  An input
    # pipeline stages a batch -- its H2D copy and, with it, engine.stage_captions -- while the previous batch is being computed;
    # the benchmark does the same INSIDE the timed region: every step first queues the length kernels + 8-byte read-back of the
    # NEXT step's captions, then runs this step on the batch staged one step earlier (same work per step, no host wait).
    # --inline-lengths: the engine stages each batch itself and waits for the read-back at the start of the step.
    staged = {"cap": None}
    # round 6: with EngineOptions.dynamic_rows (default) the row count of a packed batch never leaves the device, so there is
    # nothing to stage: the step takes the plain token tensor (the reference's API) and no host read exists anywhere in it
    dynamic = bool(eng.dynamic_rows(B, B)) and ts is None
    stage_ahead = eng.text_pack_enabled() and not args.inline_lengths and not dynamic

    def captions():
        if not stage_ahead:
            return tok
        cap = staged["cap"] or eng.stage_captions(tok)
        staged["cap"] = eng.stage_captions(tok)
        return cap

    def step():
        if ts is not None:
            loss = ts.forward(img, captions())
            ts.step(ts.backward())
            return loss
        return eng.forward_loss(img, captions(), gather=True)

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    if grouped and not hip.env_flag("MSCLIP_KEEP_DEFAULT_STREAM"):
        # with a communicator in the process the legacy default stream synchronises implicitly with RCCL's streams on every
        # launch (hip.off_default_stream): the whole run goes to one non-default stream instead of switching per call
        hip.use_compute_stream(dev)
    for _ in range(args.warmup):
        loss = step()
    if args.prefill_random:
        for v in eng._workspace(B, B).values():
            for t in (v if isinstance(v, (list, tuple)) else [v]):
                if torch.is_tensor(t) and t.is_floating_point():
                    t.normal_()
    # Per-launch durations of the dominant kernel INSIDE the timed region.  The shipped forward step is a replay of a native
    # launch table (engine "plan"): its probes are HIP timing events the table's executor records around the chosen entries on
    # their own stream (Plan.enable_probe).  The eager launch loop (training step, MSCLIP_PLAN=0) brackets the same launches from
    # the Python binding (hip.KernelProbe), as in rounds 1-5.
    planned = ts is None and eng.last_plan is not None

    def attach_probes():
        if args.no_probe:
            return None, None
        if planned:
            plan = eng.last_plan
            plan.enable_probe({"gemm:" + DOMINANT["variant"], "gemm_f8"}, args.steps)
            return ("plan", plan), None
        pr, pr8 = hip.KernelProbe(), hip.KernelProbe()                        # (pr8: the fp8 launches of PRECISION fp8 models)
        hip.set_gemm_probe(DOMINANT["variant"], pr)
        hip.set_gemm_f8_probe(pr8)
        return pr, pr8

    def detach_probes(pr, pr8):
        """-> (KernelProbe-like for the bf16 GEMM, for the fp8 GEMM); after the fence."""
        if pr is None:
            return None, None
        if isinstance(pr, tuple):
            res = pr[1].probe_results()
            return (hip.PlanProbeResults([r[1:] for r in res if r[0] != "gemm_f8"]),
                    hip.PlanProbeResults([r[1:] for r in res if r[0] == "gemm_f8"]))
        hip.set_gemm_probe(DOMINANT["variant"], None)
        hip.set_gemm_f8_probe(None)
        pr.resolve_rows()
        pr8.resolve_rows()
        return pr, pr8

    probe, probe8 = attach_probes()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    probe, probe8 = detach_probes(probe, probe8)
    shipped_plan = eng.last_plan if ts is None else None      # the launch table the timed steps replayed (None: eager loop)
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = None
    if grouped:
        # every rank's own clock over the same barrier-fenced region: the record carries the spread, so that the first run
        # on a real multi-GPU node yields a diagnosis (a straggler, an un-overlapped collective), not only a number
        allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) / args.steps * 1e3 for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_val = float(loss)
    # The shipped schedule runs the image-only convolutional branch on a side stream UNDER the projection GEMMs (+1.6-2.2 %
    # pairs/s): a GEMM launch that shares the chip measures longer than the kernel takes alone.  The per-kernel roofline is
    # therefore taken from a second pass of the same K steps with the inline schedule (nothing concurrent with the kernel
    # being timed); the timed region's own (overlapped) figure is reported beside it.
    overlapped = ts is None and eng.opt.conv_side_stream
    probe_timed, probe8_timed, dt_probe = probe, probe8, dt
    if probe is not None and overlapped:
        shipped_opt = eng.opt
        eng.opt = eng.opt.replace(conv_side_stream=False)
        step()                                           # (records the inline schedule's launch table)
        step()
        planned = ts is None and eng.last_plan is not None
        probe, probe8 = attach_probes()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_probe = time.perf_counter() - t0
        probe, probe8 = detach_probes(probe, probe8)
        eng.opt = shipped_opt

    if rank == 0:
        ms = dt / args.steps * 1e3
        pairs_s = B * world * args.steps / dt
        gf_ref = GFLOP_PER_PAIR[args.model]
        skipped = 0.0 if (ts is not None or eng.opt.full_last_block) else \
            SKIPPED_ROWS_PER_PAIR[args.model] * 18 * WIDTH[args.model] ** 2 / 1e9
        if skipped and model.precision != "fp8-qkv" and not eng.opt.last_block_all_queries:
            # ... and the last block's query projection (2 d^2 per skipped row) and attention (4 L d per skipped query) of
            # the rows that are not read afterwards (engine._last_block_attention)
            lv = SKIPPED_ROWS_PER_PAIR[args.model] + 2 - 77
            skipped += (SKIPPED_ROWS_PER_PAIR[args.model] * 2 * WIDTH[args.model] ** 2 +
                        ((lv - 1) * lv + 76 * 77) * 4 * WIDTH[args.model]) / 1e9
        # Packed captions (engine.text_pack_enabled): the rows behind a caption's EOT position do not exist.  Per text layer a
        # caption with n live rows executes 24 d^2 n (projections) + 4 d n^2 (attention over n keys) instead of the reference's
        # 24 d^2 77 + 4 d 77^2; in the compact last block only the k | v projection (4 d^2 per row) and one query's attention
        # (4 d per key) ran on every row.  Counted from this rank's actual caption lengths.
        packed = eng.text_pack_enabled()
        dead = 0.0
        if packed:
            d_, nl, Lt = WIDTH[args.model], eng.n_layers, eng.Lt
            n = lens_host.double()
            full_layers = nl if skipped == 0.0 else nl - 1
            dead = float((full_layers * ((Lt - n) * 24 * d_ * d_ + 4 * d_ * (Lt * Lt - n * n))).mean()) / 1e9
            if skipped:
                dead += float(((Lt - n) * (4 * d_ * d_ + 4 * d_)).mean()) / 1e9
        gf = gf_ref - skipped - dead               # executed algorithmic FLOPs per pair
        fmul = 3 if ts is not None else 1          # backward counted as 2x forward
        rec = {
            "metric": {"b32": "image-text pairs/sec ViT-B/32 bf16", "b16": "image-text pairs/sec ViT-B/16 bf16",
                       "l16": "image-text pairs/sec ViT-L/16-width model with the MS-CLIP-S conv stem / adapters (second C5 model), "
                              + ("bf16" if model.precision == "bf16" else "fp8 + bf16"),
                       "l14": "image-text pairs/sec MS-CLIP ViT-L/14 (patch-conv stem, 257 tokens; BASELINE config C5), "
                              + ("bf16" if model.precision == "bf16" else "fp8 + bf16")}[args.model[:3]],
            "mfma_util_pct": None,
            "value": round(pairs_s, 1), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if model.precision == "bf16" else
                     f"fp8 e4m3 operands ({'in_proj, ' if model.precision == 'fp8-qkv' else ''}c_fc, c_proj) + bf16 (everything else), fp32 accumulation",
            "data": "synthetic",
            "config": {"workload": f"MS-CLIP-S {args.model} fwd + contrastive step (both towers, gather, logits, "
                                   f"symmetric CE), per-GPU batch {B}, 224x224 images + 77-token captions, "
                                   f"random-init weights", "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "bn": "eval (folded running statistics)",
                       "captions": (f"[SOT, {args.caption_tokens} ids, EOT, zero pad]" if args.caption_tokens else
                                    "[SOT, l ~ U{4..60} random ids, EOT, zero pad] (SURVEY s8(d), synth.synth_tokens)") +
                                   f": {float(lens_host.double().mean()):.2f} live rows of {eng.Lt} per caption on average (max {int(lens_host.max())})",
                       "caption_lengths": ("computed inside the step and kept on the device (msclip_text_lengths' dims -> M_dev of every "
                                           "launch over the text rows): no host read, plain token tensors in" if dynamic else
                                           "staged one step ahead inside the timed region (engine.stage_captions: length kernels + "
                                           "8-byte read-back of step k + 1 queued in front of step k)" if stage_ahead else
                                           "computed at the start of each step (host waits for the read-back)" if packed else "not needed"),
                       "text_rows": ("packed: only the rows up to each caption's EOT position exist (the causal mask makes the rest "
                                     "unreachable from every output); MSCLIP_TEXT_PACK=0 computes all 77" if packed else
                                     "all 77 rows of every caption computed (MSCLIP_TEXT_PACK=0)")},
            "step_tflops_per_gpu": round(pairs_s / world * gf / 1e3 * fmul, 1),
            "whole_step_mfma_frac": round(pairs_s / world * gf / 1e3 * fmul / PEAK_BF16_TFLOPS, 4),   # (fp8 models: re-stated against the mixed peak below)
            "gflop_per_pair": {"reference_forward": gf_ref, "executed": round(gf * fmul, 3),
                               "not_executed_dead_rows_of_last_block": round(skipped * fmul, 3),
                               "not_executed_rows_behind_eot": round(dead * fmul, 3)},
            "loss": round(loss_val, 5),
            "launch_loop": ({"kind": "native launch table (msclip_plan_run), recorded on the first step", "entries": shipped_plan.n_ops,
                             "launches": shipped_plan.n_launches, "cross_stream_events": shipped_plan.n_events}
                            if shipped_plan is not None else {"kind": "Python / ctypes, one call per launch"}),
        }
        if ts is not None:
            rec["metric"] = f"training step (forward + backward + AdamW, BatchNorm: {args.bn} statistics) pairs/sec " + args.model
            rec["config"]["workload"] = ("forward (activations kept) + backward of every parameter (heads, loss, all "
                                         "transformer blocks, adapters, conv stem, parallel conv branch, embeddings) + AdamW + "
                                         "bucketed gradient all-reduce at N > 1; FLOPs counted as 3x forward")
            rec["config"]["bn"] = ("train mode: per-GPU batch statistics, running statistics updated (momentum 0.1)"
                                   if args.bn == "batch" else "frozen running statistics (folded); gamma / beta receive gradients")
        if not shared:
            rec["config"]["rccl_ranks"] = dist.get_world_size() if world > 1 else 1
        else:
            rec["config"]["gloo_ranks"] = dist.get_world_size()
        if per_rank is not None:
            rec["config"]["per_rank_ms_per_step"] = [round(x, 3) for x in per_rank]
            rec["config"]["rank_spread_pct"] = round(100.0 * (max(per_rank) - min(per_rank)) / max(per_rank), 2)
        if grouped and world == 1:
            rec["config"]["collectives"] = f"one-rank {dist.get_backend()} group: every collective issued (identity)"
        if shared:
            rec["config"]["TEST_ONLY"] = "all ranks share GPU 0 over gloo (MSCLIP_TEST_SHARED_GPU): not a measurement"
        n = probe.summary()[0] if probe is not None else 0
        if n > 0:                       # tiny batches never reach the ping-pong kernel: no roofline line then
            # `probe_timed` bracketed the dominant kernel's launches INSIDE the timed region (side streams on: other kernels share
            # the chip with them): that is the roofline of the number in `value`.  `probe` = the same K steps re-run with the
            # inline schedule (nothing concurrent with the launch being timed): the kernel's own rate, under roofline.isolated.
            nt, kmst, flopst = probe_timed.summary()
            alg_bytes = sum(probe_timed.bytes) / nt
            acht = flopst / (kmst * 1e-3) / 1e12
            rec["roofline"] = {"kernel": f"{DOMINANT['kernel']}, *> ({DOMINANT['what']}; incl. the LayerNorm-fold instantiations <0, false, 1 / 2> of the same main loop)",
                               "bound": "mfma", "achieved": round(acht, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(acht / PEAK_BF16_TFLOPS, 4), "traffic": None,
                               "measured_in": "the timed region (HIP events around every launch of the kernel on its launch stream)",
                               "launches_per_step": nt // args.steps, "avg_launch_us": round(kmst / nt * 1e3, 2),
                               "flops_per_launch_avg": round(flopst / nt / 1e9, 3), "flops_unit": "GFLOP",
                               "algorithmic_bytes_per_launch": round(alg_bytes),
                               "time_share_of_step": round(kmst / (dt * 1e3), 4)}
            if probe_timed is not probe:
                n, kms, flops = probe.summary()
                ach = flops / (kms * 1e-3) / 1e12
                rec["roofline"]["isolated"] = {
                    "achieved": round(ach, 1), "frac": round(ach / PEAK_BF16_TFLOPS, 4), "avg_launch_us": round(kms / n * 1e3, 2),
                    "time_share_of_step": round(kms / (dt_probe * 1e3), 4),
                    "note": (f"probe pass: the same {args.steps} steps with the conv branch inline (MSCLIP_CONV_SIDE_STREAM=0, "
                             f"{dt_probe / args.steps * 1e3:.3f} ms/step): no other kernel shares the chip with the launch being timed; "
                             "the rocprofv3 --kernel-trace summaries under profiles/ are of this schedule")}
            if args.shapes:
                rec["roofline"]["shapes"] = [
                    {"M": t[0], "N": t[1], "K": t[2], "conv": t[4], "act": t[5], "resid": t[6], "out_fp32": t[7],
                     "launches_per_step": c // args.steps, "avg_us": round(ms / c * 1e3, 1),
                     "tflops": round(u / (ms * 1e-3) / 1e12, 1), "algorithmic_MB": round(b / c / 1e6, 1)}
                    for t, (c, ms, u, b) in sorted(probe.by_shape().items(), key=lambda kv: -kv[1][1])]
            if world == 1 and not args.no_pmc:
                pmc = pmc_passes(args, DOMINANT["kernel"])
                if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                    traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
                    rec["roofline"]["traffic"] = round(traffic)
                    rec["roofline"]["traffic_unit"] = ("HBM bytes/launch, live rocprofv3 --pmc passes of this command: "
                                                       "FETCH_SIZE KiB x2 (gfx950 correction) + WRITE_SIZE KiB")
                    rec["roofline"]["traffic_over_algorithmic"] = round(traffic / alg_bytes, 3)
                if pmc and "SQ_VALU_MFMA_BUSY_CYCLES" in pmc and pmc.get("GRBM_GUI_ACTIVE"):
                    rec["roofline"]["mfma_busy_pct"] = round(100.0 * (pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD) /
                                                             (pmc["GRBM_GUI_ACTIVE"] / N_XCD), 1)
                if pmc and pmc.get("ALL_GRBM_GUI_ACTIVE"):
                    rec["roofline"]["mfma_busy_pct_whole_step"] = round(
                        100.0 * (pmc["ALL_SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD) / (pmc["ALL_GRBM_GUI_ACTIVE"] / N_XCD), 1)
                if pmc:
                    errs = {k: v for k, v in pmc.items() if k.startswith("error_")}
                    if errs:
                        rec["roofline"]["pmc_errors"] = errs
        n8 = probe8_timed.summary()[0] if probe8_timed is not None else 0
        if n8 > 0:                      # PRECISION fp8: the config's own kernel gets the headline roofline, the bf16 GEMM moves beside it
            n8, kms8, flops8 = probe8_timed.summary()
            ach8 = flops8 / (kms8 * 1e-3) / 1e12
            r8 = {"kernel": "gemm_pp_kernel<0, true> (dense fp8 e4m3 MX-MFMA GEMM: c_fc / c_proj of every block, in_proj under fp8-qkv)", "bound": "mfma",
                  "achieved": round(ach8, 1), "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s", "frac": round(ach8 / PEAK_FP8_TFLOPS, 4),
                  "traffic": None, "measured_in": "the timed region (HIP events around every launch)",
                  "launches_per_step": n8 // args.steps, "avg_launch_us": round(kms8 / n8 * 1e3, 2),
                  "flops_per_launch_avg": round(flops8 / n8 / 1e9, 3), "flops_unit": "GFLOP",
                  "algorithmic_bytes_per_launch": round(sum(probe8_timed.bytes) / n8),
                  "time_share_of_step": round(kms8 / (dt * 1e3), 4)}
            if probe8 is not probe8_timed and probe8.summary()[0] > 0:
                ni, kmsi, flopsi = probe8.summary()
                r8["isolated"] = {"achieved": round(flopsi / (kmsi * 1e-3) / 1e12, 1),
                                  "frac": round(flopsi / (kmsi * 1e-3) / 1e12 / PEAK_FP8_TFLOPS, 4),
                                  "avg_launch_us": round(kmsi / ni * 1e3, 2), "note": "inline-schedule probe pass (nothing concurrent)"}
            if "roofline" in rec:
                rec["roofline_bf16_gemm"] = rec["roofline"]
            rec["roofline"] = r8
            # whole-step fraction against the peak of the units the step's FLOPs run on: time at peak = fp8 FLOPs / 5 PF +
            # everything else / 2.5 PF, over the measured step time
            f8_step = flops8 / args.steps
            tot_step = B * gf * 1e9 * fmul
            t_peak = f8_step / (PEAK_FP8_TFLOPS * 1e12) + max(tot_step - f8_step, 0.0) / (PEAK_BF16_TFLOPS * 1e12)
            rec["whole_step_mfma_frac"] = round(t_peak / (ms * 1e-3), 4)
            rec["whole_step_mfma_frac_note"] = (f"mixed peak: {f8_step / 1e12:.2f} TFLOP/step on the fp8 MFMA (5 PF), "
                                                f"{(tot_step - f8_step) / 1e12:.2f} on bf16 (2.5 PF)")
        for key in ("roofline", "roofline_bf16_gemm"):
            if key in rec and "mfma_busy_pct_whole_step" in rec[key]:
                rec["mfma_util_pct"] = rec[key]["mfma_busy_pct_whole_step"]      # BASELINE metric's second half (all kernels of a step)
        if world == 1 and ts is None and not args.no_hbm_kernels and not args.no_pmc:
            wsp = eng._workspace(B, B)
            live = int(lens_host.sum()) if eng.text_pack_enabled() else None      # (device-side row counts: the host copy of the lengths says it)
            rec["hbm_bound_kernels"] = hbm_kernel_rates(args, B, WIDTH[args.model], eng.Lv, eng.Lt, eng.g,
                                                        Mt_live=live, Mt_rows=wsp.get("Mt"))
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args.model, sd)
        if grouped:                     # RCCL's version banner sits in libc's stdout buffer: out with it BEFORE the record, so
            import ctypes               # that the JSON line is the last line of the run
            ctypes.CDLL(None).fflush(None)
        print(json.dumps(rec), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
