// msclip_plan_*: the native step executor (plan.h has the design).  Host code only.
#include "plan.h"

#include <stdio.h>

#include "common.h"

namespace msclip_plan {
thread_local Plan* g_rec = nullptr;
thread_local int g_depth = 0;

void append(Plan* p, const char* name, void* stream, std::function<int(const Ctx&, void*)> fn) {
  const int slot = p->slot_of(stream);
  if (slot < 0) {
    if (!p->bad) p->bad_what = name;
    p->bad = true;
    return;
  }
  Op op;
  op.kind = OP_LAUNCH;
  op.slot = slot;
  op.event = -1;
  op.name = name;
  op.fn = std::move(fn);
  p->ops.push_back(std::move(op));
  ++p->n_launch;
}
}  // namespace msclip_plan

using msclip_plan::Plan;

struct msclip_plan_s { Plan p; };

extern "C" int msclip_plan_create(msclip_plan_s** out) {
  if (!out) return MSCLIP_EINVAL;
  *out = new (std::nothrow) msclip_plan_s();
  return *out ? MSCLIP_OK : MSCLIP_ELAUNCH;
}

extern "C" int msclip_plan_destroy(msclip_plan_s* h) {
  if (!h) return MSCLIP_EINVAL;
  if (msclip_plan::g_rec == &h->p) msclip_plan::g_rec = nullptr;
  for (hipEvent_t e : h->p.events) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->p.probe_ev) (void)hipEventDestroy(e);
  delete h;
  return MSCLIP_OK;
}

extern "C" int msclip_plan_begin(msclip_plan_s* h, void* const* streams, int nstreams) {
  if (!h || !streams || nstreams <= 0 || nstreams > 16 || msclip_plan::g_rec || h->p.recording || !h->p.ops.empty()) return MSCLIP_EINVAL;
  h->p.rec_streams.assign(streams, streams + nstreams);
  h->p.recording = true;
  msclip_plan::g_rec = &h->p;
  msclip_plan::g_depth = 0;
  return MSCLIP_OK;
}

extern "C" int msclip_plan_bind_external(msclip_plan_s* h, const void* base, long long nbytes) {
  if (!h || !h->p.recording || !base || nbytes <= 0 || h->p.ext.size() >= 8) return MSCLIP_EINVAL;
  h->p.ext.push_back(msclip_plan::ExtRange{(const char*)base, nbytes});
  return (int)h->p.ext.size() - 1;
}

extern "C" int msclip_plan_record_event(msclip_plan_s* h, void* stream) {
  if (!h || !h->p.recording) return MSCLIP_EINVAL;
  const int slot = h->p.slot_of(stream);
  if (slot < 0) return MSCLIP_EINVAL;
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return MSCLIP_ELAUNCH;
  h->p.events.push_back(ev);
  msclip_plan::Op op;
  op.kind = msclip_plan::OP_RECORD;
  op.slot = slot;
  op.event = (int)h->p.events.size() - 1;
  op.name = "event_record";
  h->p.ops.push_back(std::move(op));
  return op.event;
}

extern "C" int msclip_plan_wait_event(msclip_plan_s* h, void* stream, int event) {
  if (!h || !h->p.recording || event < 0 || event >= (int)h->p.events.size()) return MSCLIP_EINVAL;
  const int slot = h->p.slot_of(stream);
  if (slot < 0) return MSCLIP_EINVAL;
  msclip_plan::Op op;
  op.kind = msclip_plan::OP_WAIT;
  op.slot = slot;
  op.event = event;
  op.name = "event_wait";
  h->p.ops.push_back(std::move(op));
  return MSCLIP_OK;
}

extern "C" int msclip_plan_end(msclip_plan_s* h) {
  if (!h || !h->p.recording || msclip_plan::g_rec != &h->p) return MSCLIP_EINVAL;
  msclip_plan::g_rec = nullptr;
  h->p.recording = false;
  if (h->p.bad) {
    fprintf(stderr, "msclip_plan_end: %s was launched on a stream that is not one of the plan's stream slots\n",
            h->p.bad_what ? h->p.bad_what : "?");
    return MSCLIP_EINVAL;
  }
  return (int)h->p.ops.size();
}

// Abandon a recording (an exception in the recording pass): the plan stays empty and must be destroyed.
extern "C" int msclip_plan_abort(msclip_plan_s* h) {
  if (!h) return MSCLIP_EINVAL;
  if (msclip_plan::g_rec == &h->p) msclip_plan::g_rec = nullptr;
  h->p.recording = false;
  h->p.bad = true;
  return MSCLIP_OK;
}

extern "C" int msclip_plan_info(const msclip_plan_s* h, int* n_ops, int* n_launches, int* n_events, int* n_streams, int* n_ext) {
  if (!h) return MSCLIP_EINVAL;
  if (n_ops) *n_ops = (int)h->p.ops.size();
  if (n_launches) *n_launches = h->p.n_launch;
  if (n_events) *n_events = (int)h->p.events.size();
  if (n_streams) *n_streams = (int)h->p.rec_streams.size();
  if (n_ext) *n_ext = (int)h->p.ext.size();
  return MSCLIP_OK;
}

extern "C" const char* msclip_plan_op_name(const msclip_plan_s* h, int i) {
  if (!h || i < 0 || i >= (int)h->p.ops.size()) return nullptr;
  return h->p.ops[i].name;
}

extern "C" int msclip_plan_size(const msclip_plan_s* h) { return h ? (int)h->p.ops.size() : MSCLIP_EINVAL; }

// Launch probes: HIP timing events around the table entries op_idx[0..n) (on each entry's own stream) for the next `runs`
// replays -- the per-kernel durations bench.py's roofline leg needs, measured inside the timed region without leaving the
// native launch loop.  msclip_plan_probe_elapsed(run, i) after a device synchronise.
extern "C" int msclip_plan_probe_disable(msclip_plan_s* h) {
  if (!h) return MSCLIP_EINVAL;
  for (hipEvent_t e : h->p.probe_ev) (void)hipEventDestroy(e);
  h->p.probe_ev.clear();
  h->p.probe_slot.clear();
  h->p.probe_n = h->p.probe_runs = h->p.probe_done = 0;
  return MSCLIP_OK;
}

extern "C" int msclip_plan_probe_enable(msclip_plan_s* h, const int* op_idx, int n, int runs) {
  if (!h || h->p.recording || !op_idx || n <= 0 || runs <= 0 || (long long)n * runs > (1 << 20)) return MSCLIP_EINVAL;
  msclip_plan_probe_disable(h);
  h->p.probe_slot.assign(h->p.ops.size(), -1);
  for (int i = 0; i < n; ++i) {
    if (op_idx[i] < 0 || op_idx[i] >= (int)h->p.ops.size() || h->p.ops[op_idx[i]].kind != msclip_plan::OP_LAUNCH) return MSCLIP_EINVAL;
    h->p.probe_slot[op_idx[i]] = i;
  }
  h->p.probe_ev.resize((size_t)n * runs * 2);
  for (hipEvent_t& e : h->p.probe_ev)
    if (hipEventCreate(&e) != hipSuccess) return MSCLIP_ELAUNCH;
  h->p.probe_n = n;
  h->p.probe_runs = runs;
  h->p.probe_done = 0;
  return MSCLIP_OK;
}

extern "C" int msclip_plan_probe_runs(const msclip_plan_s* h) { return h ? h->p.probe_done : MSCLIP_EINVAL; }

extern "C" int msclip_plan_probe_elapsed(msclip_plan_s* h, int run, int i, float* ms) {
  if (!h || !ms || run < 0 || run >= h->p.probe_done || i < 0 || i >= h->p.probe_n) return MSCLIP_EINVAL;
  const size_t base = ((size_t)run * h->p.probe_n + i) * 2;
  return hipEventElapsedTime(ms, h->p.probe_ev[base], h->p.probe_ev[base + 1]) == hipSuccess ? MSCLIP_OK : MSCLIP_ELAUNCH;
}

extern "C" int msclip_plan_run(msclip_plan_s* h, void* const* streams, int nstreams, const void* const* ext, int next) {
  if (!h || h->p.recording || h->p.bad || h->p.ops.empty() || !streams || nstreams != (int)h->p.rec_streams.size() ||
      next != (int)h->p.ext.size() || (next && !ext) || msclip_plan::g_rec)
    return MSCLIP_EINVAL;
  msclip_plan::Ctx c{streams, nstreams, h->p.ext.data(), ext, next};
  const bool probing = h->p.probe_n > 0 && h->p.probe_done < h->p.probe_runs;
  hipEvent_t* pev = probing ? h->p.probe_ev.data() + (size_t)h->p.probe_done * h->p.probe_n * 2 : nullptr;
  size_t oi = 0;
  for (const msclip_plan::Op& op : h->p.ops) {
    void* st = streams[op.slot];
    const int ps = probing ? h->p.probe_slot[oi] : -1;
    ++oi;
    switch (op.kind) {
      case msclip_plan::OP_LAUNCH: {
        if (ps >= 0) (void)hipEventRecord(pev[2 * ps], (hipStream_t)st);
        const int rc = op.fn(c, st);
        if (ps >= 0) (void)hipEventRecord(pev[2 * ps + 1], (hipStream_t)st);
        if (rc != MSCLIP_OK) {
          fprintf(stderr, "msclip_plan_run: %s returned %d\n", op.name, rc);
          return rc;
        }
        break;
      }
      case msclip_plan::OP_RECORD:
        if (hipEventRecord(h->p.events[op.event], (hipStream_t)st) != hipSuccess) return MSCLIP_ELAUNCH;
        break;
      case msclip_plan::OP_WAIT:
        if (hipStreamWaitEvent((hipStream_t)st, h->p.events[op.event], 0) != hipSuccess) return MSCLIP_ELAUNCH;
        break;
      default: return MSCLIP_EINVAL;
    }
  }
  if (probing) ++h->p.probe_done;
  return MSCLIP_OK;
}

// A stream restricted to `n_cus` compute units spread evenly over the XCDs (hipExtStreamCreateWithCUMask): the side stream of the
// HBM-bound conv branch, so that its workgroups do not displace the persistent GEMM workgroups of the main stream.  CU i of the
// mask word order is (XCD i % 8, CU i / 8) on gfx950's interleaved enumeration, so the first n bits ARE an even spread.
extern "C" int msclip_stream_create_cu_masked(int n_cus, int from_top, void** stream) {
  if (!stream || n_cus <= 0) return MSCLIP_EINVAL;
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return MSCLIP_ELAUNCH;
  const int total = p.multiProcessorCount;
  if (n_cus > total) n_cus = total;
  const int words = (total + 31) / 32;
  std::vector<uint32_t> mask((size_t)words, 0u);
  for (int i = 0; i < n_cus; ++i) {
    const int bit = from_top ? total - 1 - i : i;
    mask[(size_t)bit / 32] |= 1u << (bit % 32);
  }
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()) != hipSuccess) return MSCLIP_ELAUNCH;
  *stream = (void*)s;
  return MSCLIP_OK;
}
