// Launch plans: record the launches of one step ONCE, replay them from a native table (include/msclip_hip.h, msclip_plan_*).
//
// Why: the engine's step is ~230 (forward) to ~900 (training) launches issued from Python through ctypes -- 15-40 us of host time
// each (descriptor building, argument marshalling, stream lookups).  Everything a launch needs is known after the first step
// with a given batch shape: buffers live in a persistent workspace, row counts that depend on the data (packed captions) are read
// by the kernels from device memory (msclip_gemm_desc.M_dev and friends).  So the step is recorded while it runs -- every exported
// entry point, when a plan is recording on the calling thread, appends (its own address, a copy of its arguments, the stream's
// slot) to the plan before doing its work -- and replayed by msclip_plan_run: a loop over that table calling the same entry
// points with the stored arguments, ~2-3 us per launch, no Python, no allocation.  Cross-stream edges (side streams for the conv
// branch / text block 0) are recorded as event record / wait entries on stream slots; the caller passes the actual streams at run
// time.  Pointers into buffers that change from call to call (the input images, the token ids) are registered as "externals" while
// recording and re-based at run time.  A replay under hipStreamBeginCapture is an ordinary stream capture (kernel launches, event
// records and waits only), which is how engine.graph() captures the packed step.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/msclip_hip.h"

namespace msclip_plan {

struct ExtRange {
  const char* base;     // address range the recording saw
  long long nbytes;
};

struct Ctx {            // one replay
  void* const* streams;
  int nstreams;
  const ExtRange* ext;  // recorded ranges
  const void* const* now;   // this call's base pointers (same order)
  int next;
  const void* fix(const void* p) const {
    const char* c = (const char*)p;
    for (int i = 0; i < next; ++i)
      if (c >= ext[i].base && c < ext[i].base + ext[i].nbytes) return (const char*)now[i] + (c - ext[i].base);
    return p;
  }
};

enum OpKind { OP_LAUNCH = 0, OP_RECORD = 1, OP_WAIT = 2 };

struct Op {
  int kind;
  int slot;             // stream slot
  int event;            // OP_RECORD / OP_WAIT
  const char* name;     // entry point (static string)
  std::function<int(const Ctx&, void*)> fn;
};

struct Plan {
  std::vector<Op> ops;
  std::vector<void*> rec_streams;      // handles registered for the recording: index = slot
  std::vector<ExtRange> ext;
  std::vector<hipEvent_t> events;
  bool recording = false;
  // launch probes (msclip_plan_probe_*): timing events around chosen table entries for the first `probe_runs` replays
  std::vector<int> probe_slot;         // per op: index into the probed set, or -1
  std::vector<hipEvent_t> probe_ev;    // [run][probed op][2]
  int probe_n = 0, probe_runs = 0, probe_done = 0;
  bool bad = false;                    // a launch on a stream that was not registered, or a failed entry point
  const char* bad_what = nullptr;
  int n_launch = 0;
  int slot_of(void* stream) {
    for (size_t i = 0; i < rec_streams.size(); ++i)
      if (rec_streams[i] == stream) return (int)i;
    return -1;
  }
};

extern thread_local Plan* g_rec;       // the plan recording on this thread (nullptr: none)
extern thread_local int g_depth;       // entry points that call other entry points record only the outermost one

struct Scope {
  bool on;
  explicit Scope(bool a) : on(a) { if (on) ++g_depth; }
  ~Scope() { if (on) --g_depth; }
};

// ---- argument storage: descriptors passed by pointer are copied by value; everything else is stored as it is
struct GemmDescVal { msclip_gemm_desc d; };
struct QkvDescVal { msclip_qkvattn_desc d; };
template <class T> inline T wrap(T v) { return v; }
inline GemmDescVal wrap(const msclip_gemm_desc* d) { return GemmDescVal{*d}; }
inline QkvDescVal wrap(const msclip_qkvattn_desc* d) { return QkvDescVal{*d}; }

// ---- argument materialisation at replay: pointers that fall into a registered external range are re-based
template <class T> inline T fix_arg(const Ctx& c, T& v) {
  if constexpr (std::is_pointer_v<T>) return (T)c.fix((const void*)v);
  else return v;
}
inline const msclip_gemm_desc* fix_arg(const Ctx& c, GemmDescVal& v) {
  if (c.next) v.d.X = c.fix(v.d.X);
  return &v.d;
}
inline const msclip_qkvattn_desc* fix_arg(const Ctx& c, QkvDescVal& v) {
  if (c.next) v.d.X = c.fix(v.d.X);
  return &v.d;
}

void append(Plan* p, const char* name, void* stream, std::function<int(const Ctx&, void*)> fn);

template <class F, class... A>
inline void record(const char* name, F f, void* stream, A... a) {
  auto tup = std::make_tuple(wrap(a)...);
  append(g_rec, name, stream, [f, tup](const Ctx& c, void* st) -> int {
    auto t = tup;                                           // (descriptor copies are patched in place: work on a copy)
    return std::apply([&](auto&... x) -> int { return f(fix_arg(c, x)..., st); }, t);
  });
}

}  // namespace msclip_plan

// First statement of every stream-ordered entry point: fn(args..., stream).  Records the call when a plan is recording on this
// thread, then the entry point runs as usual (the recording pass IS a real step).
// Entry points whose arguments include HOST arrays (read before the call returns) cannot be replayed from a stored pointer: a
// recording that reaches one is marked unusable, loudly (msclip_plan_end fails naming the entry point).
#define MSCLIP_PLAN_UNSUPPORTED(fn)                                            \
  if (msclip_plan::g_rec && msclip_plan::g_depth == 0 && !msclip_plan::g_rec->bad) {  \
    msclip_plan::g_rec->bad = true;                                            \
    msclip_plan::g_rec->bad_what = #fn " (host-array arguments: not recordable)";      \
  }
#define MSCLIP_PLAN_HOOK(fn, stream, ...)                                      \
  msclip_plan::Scope plan_scope_(msclip_plan::g_rec != nullptr);               \
  if (msclip_plan::g_rec && msclip_plan::g_depth == 1) msclip_plan::record(#fn, fn, stream, __VA_ARGS__)
