"""Gradient GEMMs of the training step on the forward's MFMA kernels (msclip_gemm / msclip_gemm_splitk):
dW = dY^T . X (`wgrad`: both operands transposed so that the token / pixel axis is the GEMM's K axis, the deep contraction
cut into slices) and dX = dY . W (`dgrad`: the transposed weight as the W operand).  Used by train.py (transformer
blocks, heads) and train_conv.py (convolutions through im2col / col2im)."""
import torch

from . import hip
from . import options

BF = torch.bfloat16
F32 = torch.float32


def _tn_ok(dy_bf, x_bf, ragged=False):
    """The token-major split-K GEMM (msclip_gemm_splitk_tn: no operand transposes).  The wide gradients (transformer
    projections) are whole 256-channel tiles on both sides; ragged=True admits any channel counts (the conv side's 48-192
    channels: an edge tile's extra channels only reach outputs that are not stored).  Operands that do not qualify (odd strides /
    channel counts) take the transposing path."""
    if not (dy_bf.stride(1) == 1 and x_bf.stride(1) == 1 and dy_bf.stride(0) % 8 == 0 and x_bf.stride(0) % 8 == 0):
        return False
    if ragged:
        return x_bf.shape[1] % 4 == 0
    return dy_bf.shape[1] % 256 == 0 and x_bf.shape[1] % 256 == 0


def wgrad(dy_bf, x_bf, M, out=None):
    """dW[N, K] = dY^T @ X over the first M rows of dy_bf [*, N] and x_bf [*, K] (bf16): both operands transposed so the
    token axis is the contiguous K axis of the GEMM (zero-padded).  A weight gradient has few output tiles over a very
    deep contraction (65 024 tokens at batch 512, up to 6.4 M pixels): the contraction is cut into S slices, workgroup
    rows of ONE launch, into fp32 partials folded in a fixed order (deterministic).  `out` (fp32 [N, K], contiguous):
    where the gradient is written (a slot of a gradient bucket, comm.GradReducer.reserve)."""
    N, K = dy_bf.shape[1], x_bf.shape[1]
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    if tiles <= 4 and M >= 65536 and _tn_ok(dy_bf, x_bf, ragged=True):
        # narrow gradients over many pixels (the stem / parallel-branch convolutions: 0.4-1.6 M pixels at batch 512): token-major
        # too -- 256 x 256 tiles waste MFMA rows on 48-192 channels, but the job is HBM-bound and the transposes were 2/3 of its
        # traffic (1.6 M x (96 + 448): 501 us against 1116 us; below ~65 k tokens the transposing path's 128 x 128 tiles win)
        return hip.gemm_splitk_tn(dy_bf, x_bf, M, max(1, min(256 // tiles, M // 2048)), out=out)
    if tiles <= 4:
        # narrow gradients (the convolutions: 48-192 channels over up to 6.4 M pixels): one split-K launch of 128 x 128
        # tiles, enough slices to put ~2 workgroups on every CU, at least 1024 deep each
        t128 = ((N + 127) // 128) * ((K + 127) // 128)
        S = max(1, min(512 // t128, M // 1024))
        Mpad = (M + 64 * S - 1) // (64 * S) * (64 * S)
        return hip.gemm_splitk(hip.transpose_bf16(dy_bf, M, Mpad), hip.transpose_bf16(x_bf, M, Mpad), S, out=out)
    # wide gradients: ONE split-K launch of the ping-pong GEMM (tiles x slices workgroups, see WgradJob)
    S = max(1, min(256 // tiles, M // 2048))
    if _tn_ok(dy_bf, x_bf):                                # token-major operands as they are: no transposes
        return hip.gemm_splitk_tn(dy_bf, x_bf, M, S, out=out)
    Mpad = (M + 64 * S - 1) // (64 * S) * (64 * S)
    a = hip.transpose_bf16(dy_bf, M, Mpad)
    b = hip.transpose_bf16(x_bf, M, Mpad)
    if out is None:
        out = torch.empty(N, K, dtype=F32, device=a.device)
    if S == 1:
        hip.gemm(a, b, out)
        return out
    return hip.gemm_splitk(a, b, S, out=out, tile=4)


def dgrad(dy_bf, w_t, out=None):
    """dX[M, K] = dY[M, N] @ W[N, K], w_t = W^T-as-stored-for-the-GEMM = [K, N] bf16 (N % 64 == 0)."""
    if out is None:
        out = torch.empty(dy_bf.shape[0], w_t.shape[0], dtype=BF, device=dy_bf.device)
    hip.gemm(dy_bf, w_t, out)
    return out


# ---- weight gradients off the critical path.  The backward's dependency chain is the dgrad GEMMs; a weight gradient is
# only needed by the optimizer (or the gradient all-reduce).  wgrad_async runs the whole weight-gradient job -- the two
# operand transposes (HBM-bound), the split-K GEMMs and the fold -- on a side "lane" stream behind an event, so it overlaps
# the MFMA-bound dgrad GEMMs of the main stream; join() makes the current stream wait for everything queued on the lane.
_LANE = {}


def lane(device):
    s = _LANE.get(device)
    if s is None:
        s = _LANE[device] = hip.background_stream(device)
    return s


def lane_stream(device):
    """The lane stream of `device` if anything has used it yet, else None (comm.GradReducer orders its buckets behind it)."""
    return _LANE.get(device)


def _ranks_share_a_gpu():
    """gloo over CUDA tensors = the test setup where several ranks time-slice ONE GPU (tests/test_gpu_model.py): there the
    extra stream and its cross-stream events make a step 30 x slower (5.1 s against 0.16 s at batch 8; every gloo
    collective also blocks the host until the stream has drained).  One GPU per rank (nccl = RCCL) uses the lane."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo"


def wgrad_async(dy_bf, x_bf, M, post=None, out=None):
    """wgrad(dy_bf, x_bf, M) on the lane stream.  The operands must not be overwritten in place afterwards (their memory
    may be freed: the caching allocator is told about the lane's use); the result may only be touched after join().
    x_bf may be a callable that builds the operand: it runs on the lane as well."""
    dev = dy_bf.device
    if options.TRAIN.wgrad_sync or _ranks_share_a_gpu():
        out = wgrad(dy_bf, x_bf() if callable(x_bf) else x_bf, M, out)    # everything on the calling stream (A/B knob; gloo test setups)
        return post(out) if post is not None else out
    cur, ln = torch.cuda.current_stream(dev), lane(dev)
    ready = torch.cuda.Event()
    ready.record(cur)
    ln.wait_event(ready)
    with torch.cuda.stream(ln):
        if callable(x_bf):                               # the operand itself is lane work (a convolution's column matrix)
            x_bf = x_bf()
        else:
            x_bf.record_stream(ln)
        out = wgrad(dy_bf, x_bf, M, out)
        if post is not None:
            out = post(out)
    dy_bf.record_stream(ln)
    out.record_stream(cur)
    return out


class WgradJob:
    """dW[N, K] = dY^T @ X for a WIDE weight gradient (the transformer's projections: 9-36 output tiles of 256 x 256 over a
    65 024-deep contraction), in two halves:

    * now: the two operand transposes (HBM-bound) on the lane stream, behind an event of the calling stream;
    * finish(), a few kernels later on the calling stream: ONE split-K launch of the ping-pong GEMM -- tiles x slices ~ 250
      workgroups, one tile each, so the launch fills the chip and nothing of it lingers on a few CUs beside the next
      kernels -- and the fold of the fp32 partials (fixed order: deterministic).

    Why not K-slices on side streams (wgrad above, what round 2 shipped for these): a slice is a persistent GEMM of 9-36
    workgroups that holds its CUs for ~0.6 ms; the main stream's GEMMs launch 256 persistent workgroups with a STATIC tile
    list each, so while slices hold 18-72 CUs that many of them start only after another has finished its whole list --
    up to two rounds instead of one.  One full-chip launch in the main stream's own order has neither problem, and the
    lane keeps only bandwidth-bound work, which shares the chip with MFMA-bound GEMMs far better."""

    def __init__(self, dy_bf, x_bf, M, post=None):
        self.N, self.K, self.M, self.post = dy_bf.shape[1], x_bf.shape[1], M, post
        tiles = ((self.N + 255) // 256) * ((self.K + 255) // 256)
        self.S = S = max(1, min(256 // tiles, M // 2048))
        Mpad = (M + 64 * S - 1) // (64 * S) * (64 * S)
        dev = dy_bf.device
        self.sync = options.TRAIN.wgrad_sync or _ranks_share_a_gpu()
        self.tn = _tn_ok(dy_bf, x_bf)
        if self.tn:
            # round 4: the split-K GEMM reads both operands token-major (LDS transpose reads): nothing to do until finish().
            # The operands stay referenced here (same contract as before: never overwritten in place afterwards).
            self.a, self.b, self.done = dy_bf, x_bf, None
            return
        if self.sync:
            self.a, self.b, self.done = hip.transpose_bf16(dy_bf, M, Mpad), hip.transpose_bf16(x_bf, M, Mpad), None
            return
        cur, ln = torch.cuda.current_stream(dev), lane(dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        ln.wait_event(ready)
        with torch.cuda.stream(ln):
            self.a = hip.transpose_bf16(dy_bf, M, Mpad)
            self.b = hip.transpose_bf16(x_bf, M, Mpad)
            self.done = torch.cuda.Event()
            self.done.record(ln)
        dy_bf.record_stream(ln)
        x_bf.record_stream(ln)

    def finish(self, out=None):
        """-> fp32 [N, K] (written into `out` when given, e.g. a gradient bucket's slot) on the calling stream."""
        dev = self.a.device
        cur = torch.cuda.current_stream(dev)
        if self.done is not None:
            cur.wait_event(self.done)
            self.a.record_stream(cur)
            self.b.record_stream(cur)
        if out is None:
            out = torch.empty(self.N, self.K, dtype=F32, device=dev)
        if self.tn:
            hip.gemm_splitk_tn(self.a, self.b, self.M, self.S, out=out)
        elif self.S == 1:
            hip.gemm(self.a, self.b, out, tile=4)
        else:
            hip.gemm_splitk(self.a, self.b, self.S, out=out, tile=4)
        self.a = self.b = None
        return self.post(out) if self.post is not None else out


def on_lane(fn, *operands):
    """fn() on the lane stream behind an event (same contract as wgrad_async: operands read-only afterwards, result valid
    after join).  For the other HBM-bound by-products of the backward that nothing on the critical path reads (bias sums)."""
    dev = operands[0].device
    if options.TRAIN.wgrad_sync or _ranks_share_a_gpu():
        return fn()
    cur, ln = torch.cuda.current_stream(dev), lane(dev)
    ready = torch.cuda.Event()
    ready.record(cur)
    ln.wait_event(ready)
    with torch.cuda.stream(ln):
        out = fn()
    for t in operands:
        t.record_stream(ln)
    out.record_stream(cur)
    return out


def join(device):
    if device in _LANE:
        torch.cuda.current_stream(device).wait_stream(_LANE[device])
