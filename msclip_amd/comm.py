"""Rank helpers and the feature all-gather of the contrastive head.

Mirrors reference lib/utils/comm.py: `Comm` (:12-62) and `gather_tensors`
(:140-154): every rank ends up with the rank-major concatenation of all ranks'
rows, and autograd flows only through the local slice.  Differences, all
MI355X-motivated: one collective into one contiguous buffer
(`all_gather_into_tensor`, which is ncclAllGather == RCCL on ROCm) instead of
world x `ones_like` + list all_gather + cat; image and text features can be
packed into a single [B, 2, E] call (`gather_features`); the engine issues one
asynchronous gather per modality from a side HIP stream (`gather_rows_async`: the image
gather runs under the text head); and an uninitialised process group means world size 1 instead of
an exception.
"""
import torch
import torch.distributed as dist


class Comm(object):
    def __init__(self, local_rank=0):
        self._local_rank = local_rank

    @property
    def world_size(self):
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @property
    def collectives(self):
        """True when the data path has to run its collectives: more than one rank -- or ONE rank of an initialised group
        with MSCLIP_COLLECTIVES_AT_WORLD_1=1 (the one-GPU RCCL test: every collective then goes through the backend as an
        identity instead of being skipped)."""
        import os
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or os.environ.get("MSCLIP_COLLECTIVES_AT_WORLD_1") == "1"

    @property
    def rank(self):
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @property
    def local_rank(self):
        return self._local_rank if dist.is_available() and dist.is_initialized() else 0

    @local_rank.setter
    def local_rank(self, value):
        self._local_rank = value

    @property
    def head(self):
        return "Rank[{}/{}]".format(self.rank, self.world_size)

    def is_main_process(self):
        return self.rank == 0

    def synchronize(self):
        if not self.collectives:
            return
        dist.barrier()


comm = Comm()


def _all_gather_rows(t):
    world = comm.world_size
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    # ncclAllGather on RCCL; gloo implements the flat form too (torch >= 2.0).  No fallback: a failing collective surfaces.
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


def gather_tensors(tensor):
    """[B, ...] on every rank -> [B*world, ...], rank-major; gradient only through the local rows."""
    if not comm.collectives:
        return tensor
    out = _all_gather_rows(tensor.detach())
    if tensor.requires_grad:
        b, r = tensor.shape[0], comm.rank
        out = torch.cat([out[:r * b], tensor, out[(r + 1) * b:]], dim=0)
    return out


def gather_features(packed):
    """packed [B, 2, E] (image | text features of the local batch) -> [world*B, 2, E] in ONE collective."""
    if not comm.collectives:
        return packed
    return _all_gather_rows(packed)


_SIDE = {}


def side_stream(device):
    """The HIP stream the feature collectives are issued from (one per device): RCCL orders its kernel behind the
    work already queued on the stream that is current at the call, so issuing from a side stream that waited for
    "features ready" keeps the collective independent of everything the compute stream launches afterwards."""
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


class GatherHandle:
    """An in-flight all-gather.  wait() makes the CURRENT stream wait for it (the host does not block).  An asynchronous
    RCCL error (ncclCommGetAsyncError; ProcessGroupNCCL polls it in its watchdog and rethrows it from Work.wait) is
    re-raised here with the rank and the collective named, instead of surfacing later as a hang or a bare abort."""

    def __init__(self, work, out, side):
        self.work, self.out, self.side = work, out, side

    def wait(self):
        try:
            self.work.wait()
        except RuntimeError as exc:
            raise RuntimeError(f"{comm.head} feature all-gather of {tuple(self.out.shape)} failed: {exc}") from exc
        if self.side is not None:
            torch.cuda.current_stream(self.out.device).wait_stream(self.side)
        return self.out


def gather_rows_async(t, out=None):
    """Start the rank-major all-gather of t [B, ...]; returns (out, handle).  On a HIP device the collective is issued
    from the side stream behind an event recorded now on the compute stream ("t is final"), so it overlaps whatever
    the compute stream runs next.  handle is None at world size 1 (out is t itself).  `out`: a persistent [world * B, ...]
    buffer of the caller (the engine's workspace: no allocation per step); it is overwritten behind the event, i.e. after
    everything the compute stream had queued on the previous step's gathered rows."""
    if not comm.collectives:
        return t, None
    t = t.contiguous()
    shape = (comm.world_size * t.shape[0],) + tuple(t.shape[1:])
    if out is None or tuple(out.shape) != shape or out.dtype != t.dtype or out.device != t.device:
        out = torch.empty(shape, dtype=t.dtype, device=t.device)
    if not t.is_cuda:
        return out, GatherHandle(dist.all_gather_into_tensor(out, t, async_op=True), out, None)
    side = side_stream(t.device)
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(t.device))
    side.wait_event(ready)
    with torch.cuda.stream(side):
        work = dist.all_gather_into_tensor(out, t, async_op=True)
    out.record_stream(side)
    t.record_stream(side)
    return out, GatherHandle(work, out, side)


_ARENA = {}       # (device, bucket_elems) -> [persistent flat fp32 buckets]: allocated once, re-used by every step


class GradReducer:
    """Gradient averaging over the data-parallel ranks (what the reference's DDP wrapper does for its trainer), as a few
    large all-reduces that overlap the rest of the backward pass.

    The backward hands over gradients in the order it produces them (last block first).  They live in flat fp32 buckets of
    ~`bucket_bytes` that persist across steps (no per-step allocation): a producer that can write to a given address asks
    for its slot first (`reserve`) and the gradient is born inside the bucket -- the transformer's weight gradients, ~340 of
    the model's 530 MB; the remaining tensors are copied in by one multi-tensor copy per bucket.  A full bucket is all-reduced asynchronously from the side stream behind an event recorded on the compute
    stream ("bucket is final"), so RCCL's ring runs under the remaining GEMMs.  Bucket size: xGMI is point-to-point
    (7 links x ~153 GB/s per GPU) and a ring all-reduce is bound by ONE link's bandwidth, so the collectives should be few
    and tens of MB each (the ~1 ms ring latency of an 8-GPU node amortised to a few percent), not one per tensor: the
    132 M-parameter model makes eight 64 MiB buckets.  finish() waits for every collective and returns views into the
    averaged buckets under the original names; they stay valid until the next reducer of the same bucket size on this
    device starts filling its buckets (i.e. until the next backward).  World size 1: a pass-through."""

    def __init__(self, bucket_bytes=64 << 20, average=True):
        self.generation = None       # arena generation this reducer's buckets belong to (set when it takes its first bucket)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.average = average
        self.flat = None             # the open bucket
        self.fill = 0                # elements handed out in it
        self.layout = []             # [(name, shape, offset, numel)] of the open bucket
        self.copies = []             # [(view, source)] still to be copied into the open bucket
        self.reserved = {}           # name -> view handed out by reserve() and not yet add()ed
        self.used = 0                # buckets of the arena taken by this reducer
        self.flights = []            # [(flat, work, side, layout, keepalive)]
        self.launched = 0            # collectives issued (tests / bench read it)

    def _bucket(self, device, need):
        """Open bucket with room for `need` elements (a tensor larger than a bucket gets one of its own size)."""
        if self.flat is not None and self.fill + need > self.flat.numel() and self.fill:
            self.flush()
        if self.flat is None:
            size = max(self.bucket_elems, need)
            arena = _ARENA.setdefault((device, self.bucket_elems), [])
            if self.used == 0:
                # this reducer starts overwriting the arena: whatever an earlier reducer's finish() handed out is stale from here
                key = (device, self.bucket_elems)
                _ARENA_GEN[key] = _ARENA_GEN.get(key, 0) + 1
                self.generation, self.arena_key = _ARENA_GEN[key], key
            while len(arena) <= self.used:
                arena.append(None)
            if arena[self.used] is None or arena[self.used].numel() < size:
                arena[self.used] = torch.empty(size, dtype=torch.float32, device=device)
            self.flat, self.fill = arena[self.used], 0
            self.used += 1
        return self.flat

    def reserve(self, name, shape, device):
        """-> an fp32 view of `shape` inside the open bucket for a producer that writes its result in place (None at world
        size 1).  The producer's add(name, view) must follow before any other gradient is handed over."""
        if not comm.collectives:
            return None
        if self.reserved:
            raise RuntimeError(f"GradReducer.reserve({name!r}): the slot of {next(iter(self.reserved))!r} is still open "
                               "(a reserve() must be followed by its add() before the next gradient is handed over)")
        n = 1
        for d in shape:
            n *= int(d)
        flat = self._bucket(device, n)
        view = flat[self.fill:self.fill + n].view(tuple(shape))
        self.reserved[name] = (view, self.fill, n)
        return view

    def add(self, name, grad):
        if not comm.collectives:
            self.flights.append((None, None, None, [(name, grad)], None))
            return
        if self.reserved and name not in self.reserved:
            raise RuntimeError(f"GradReducer.add({name!r}) while the reserved slot of {next(iter(self.reserved))!r} is open: "
                               "it would overlap that slot")
        r = self.reserved.pop(name, None)
        if r is not None and r[0].data_ptr() == grad.data_ptr() and tuple(r[0].shape) == tuple(grad.shape):
            o, n = r[1], r[2]                                  # born in the bucket: nothing to copy
        else:
            n = grad.numel()
            flat = self._bucket(grad.device, n)
            o = self.fill
            self.copies.append((flat[o:o + n], grad.reshape(-1)))
        self.layout.append((name, tuple(grad.shape), o, n))
        self.fill = o + n
        if self.fill >= self.bucket_elems:
            self.flush()

    def flush(self):
        if self.flat is None or not self.layout:
            return
        flat, fill, layout, copies = self.flat, self.fill, self.layout, self.copies
        self.flat, self.fill, self.layout, self.copies = None, 0, [], []
        dev = flat.device
        payload = flat[:fill]

        def pack():
            if copies:
                dsts, srcs = [d for d, _ in copies], [g for _, g in copies]
                if all(g.dtype == torch.float32 for g in srcs):
                    torch._foreach_copy_(dsts, srcs)                          # one multi-tensor launch per bucket
                else:
                    for d, g in copies:
                        d.copy_(g)
            if self.average and comm.world_size > 1:
                payload.mul_(1.0 / comm.world_size)                           # pre-scaled: the reduced sum is the mean

        side = None
        if flat.is_cuda:
            # pack + all-reduce both on the side stream, behind "everything queued so far" on the compute stream AND on the
            # wgrad lane stream (gradgemm.wgrad_async): the compute stream itself waits for neither and goes on with the
            # backward pass
            from .gradgemm import lane_stream
            side = side_stream(dev)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            side.wait_event(ready)
            lane = lane_stream(dev)
            if lane is not None:
                side.wait_stream(lane)
            with torch.cuda.stream(side):
                pack()
                work = dist.all_reduce(payload, async_op=True)
            for _, g in copies:
                g.record_stream(side)
        else:
            pack()
            work = dist.all_reduce(payload, async_op=True)
        self.launched += 1
        self.flights.append((flat, work, side, layout, [g for _, g in copies]))   # sources: alive until the side stream has read them

    def finish(self, clone=False):
        """-> {name: averaged gradient}; the current stream is ordered behind every collective.  At world size > 1 the values
        are VIEWS into the per-device bucket arena: they are overwritten by the next reducer on this device (the next
        backward) -- consume them (optimizer step) before that, or ask for clone=True (accumulating over micro-batches,
        comparing two backward passes).  The returned GradViews knows its arena generation: check_fresh() raises once a later
        reducer has started filling the buckets."""
        self.flush()
        out = GradViews()
        out.generation, out.arena_key = (None, None) if clone else (self.generation, getattr(self, "arena_key", None))
        for flat, work, side, layout, _ in self.flights:
            if flat is None:
                out[layout[0][0]] = layout[0][1]
                continue
            try:
                work.wait()
            except RuntimeError as exc:
                raise RuntimeError(f"{comm.head} gradient all-reduce of a {flat.numel()}-element bucket failed: {exc}") from exc
            if side is not None:
                torch.cuda.current_stream(flat.device).wait_stream(side)
            for name, shape, o, n in layout:
                v = flat[o:o + n].view(shape)
                out[name] = v.clone() if clone else v
        self.flights = []
        return out


_ARENA_GEN = {}


class GradViews(dict):
    """GradReducer.finish()'s result: a dict whose tensors may be views into the shared bucket arena."""
    generation = None
    arena_key = None

    def check_fresh(self):
        if self.generation is not None and _ARENA_GEN.get(self.arena_key) != self.generation:
            raise RuntimeError("these gradients are views into the gradient-bucket arena and a later backward() on this device "
                               "has overwritten them: step() right after backward(), or backward(clone=True)")


def local_label_offset(local_batch):
    """Global label of local row i is offset + i (rank-major gather order, comm.py:150-153)."""
    return comm.rank * local_batch


def init_distributed(backend=None):
    """env:// initialisation used by bench.py (reference lib/utils/utils.py:61-73: nccl == RCCL on ROCm)."""
    import datetime
    import os
    if dist.is_initialized():
        return
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("MSCLIP_COLLECTIVES_AT_WORLD_1") != "1":
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        comm.local_rank = local
    dist.init_process_group(backend=backend, init_method="env://", timeout=datetime.timedelta(minutes=30))


# ---------------------------------------------------------------------------------------------------------------------
# RCCL through the library's C ABI (include/msclip_hip.h: msclip_comm_init / msclip_allgather_feats / msclip_allreduce;
# SURVEY.md s8(b)).  The collectives above go through torch.distributed's ProcessGroupNCCL -- a Python-side object with its own
# streams and watchdog; these run on the CALLER's stream, which makes them table entries of a launch plan (hip.Plan) and legal
# inside a hipGraph capture, and lets a host without Python drive the same path.  Opt-in: EngineOptions.native_collectives
# (MSCLIP_NATIVE_COLLECTIVES=1) + init_native_comm() once per process, after init_distributed().
# ---------------------------------------------------------------------------------------------------------------------
_NATIVE = {"comm": None, "world": 1, "rank": 0}


def native_comm():
    """The C-ABI communicator handle (ctypes.c_void_p) of this process, or None before init_native_comm()."""
    return _NATIVE["comm"]


def init_native_comm(device=None):
    """Collective over the ranks of the torch.distributed group (or a one-rank communicator without one): rank 0 creates the
    128-byte RCCL unique id (msclip_comm_unique_id), the existing process group's store carries it to the others
    (broadcast_object_list), every rank joins (msclip_comm_init).  Replaces nothing of the reference: its rendezvous is
    torch.distributed.init_process_group (lib/utils/utils.py:61-73), which stays the bootstrap here."""
    import ctypes
    from . import hip
    if _NATIVE["comm"] is not None:
        return _NATIVE["comm"]
    grouped = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if grouped else (0, 1)
    L = hip.lib()
    buf = ctypes.create_string_buffer(128)
    if rank == 0:
        hip._check(L.msclip_comm_unique_id(buf), "msclip_comm_unique_id")
    ident = [bytes(buf.raw)]
    if grouped and world > 1:
        dist.broadcast_object_list(ident, src=0)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        hip._check(L.msclip_comm_init(rank, world, ctypes.create_string_buffer(ident[0], 128), ctypes.byref(handle)), "msclip_comm_init")
    _NATIVE.update(comm=handle, world=world, rank=rank)
    return handle


def destroy_native_comm():
    from . import hip
    h, _NATIVE["comm"] = _NATIVE["comm"], None
    if h is not None:
        hip._check(hip.lib().msclip_comm_destroy(h), "msclip_comm_destroy")


def native_world():
    return _NATIVE["world"]


def native_rank():
    return _NATIVE["rank"]
