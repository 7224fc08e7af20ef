"""msclip_amd -- MI355X-native MS-CLIP-S contrastive hot path."""
__version__ = "0.1.0"

import os as _os

# The HIP runtime hands out kernel-argument memory from a ring of four chunks and blocks a launch that enters a chunk until
# everything launched from it one ring ago has retired: with the default ring the host's lead over the GPU is capped at a few
# hundred launches wherever launches carry large argument blocks (ATen's TensorIterator kernels, the AdamW tensor tables), which
# is exactly where the training step's kernels are short -- the GPU then waits for the host at every step boundary
# (tools/probes/queue_depth_probe.py).  A 16 MiB ring removes the cap (training step -1.3 % / -2.7 %, forward
# unchanged).  Read by the runtime at its first HIP call: effective when this package (or the variable) comes before that.
_os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(16 << 20))
