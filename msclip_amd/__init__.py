"""msclip_amd -- MI355X-native MS-CLIP-S contrastive hot path.

Importing this package changes nothing in the process.  The two process-wide settings the benchmark and the training tools use
are explicit calls (INTEGRATION.md "Runtime configuration"):

* msclip_amd.configure_runtime()  -- HSA_KERNARG_POOL_SIZE, BEFORE the first HIP call of the process;
* msclip_amd.hip.use_compute_stream(device)  -- leave the legacy default stream once a process group exists.
"""
__version__ = "0.1.0"

import os as _os


def configure_runtime(kernarg_pool_bytes=16 << 20):
    """Process-wide HIP runtime settings this library's training step was tuned with; call BEFORE the first HIP call of the
    process (before anything touches the GPU).  Returns {name: value actually in effect}.

    HSA_KERNARG_POOL_SIZE: the HIP runtime hands out kernel-argument memory from a ring of four chunks and blocks a launch that
    enters a chunk until everything launched from it one ring ago has retired.  With the default ring the host's lead over the
    GPU is capped at a few hundred launches wherever launches carry large argument blocks (ATen's TensorIterator kernels, the
    AdamW tensor tables) -- where the training step's kernels are short, so the GPU waits for the host at every step boundary
    (tools/probes/queue_depth_probe.py).  A 16 MiB ring removes the cap: training step -1.3 % / -2.7 %, forward
    unchanged.  A value the caller already exported is kept."""
    _os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(int(kernarg_pool_bytes)))
    try:
        import torch
        late = torch.cuda.is_initialized()
    except Exception:
        late = False
    if late:
        import warnings
        warnings.warn("msclip_amd.configure_runtime(): the HIP runtime is already initialised in this process; "
                      "HSA_KERNARG_POOL_SIZE only takes effect when set before the first HIP call")
    return {"HSA_KERNARG_POOL_SIZE": _os.environ["HSA_KERNARG_POOL_SIZE"], "effective": not late}
