// Internal host-side plumbing shared by the libh2mi.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/h2mi.h"

namespace h2 {

struct ProfRec {
  std::string name;
  hipEvent_t a, b;
};

// One entry per device the process drives (h2mi_init: one; h2mi_init_devices(n): n).  Entry 0 is the primary
// device: transforms, polynomial helpers and the final fold of a sharded MSM run there.  With fewer physical GPUs
// than requested (H2MI_VIRTUAL_DEVICES=1) several entries share one GPU — a rehearsal mode for one-GPU boxes.
struct DevCtx {
  int device = -1;
  hipStream_t stream = nullptr, head_stream = nullptr, accum_stream = nullptr, tail_stream = nullptr;
};

struct Ctx {
  bool inited = false;
  int device = -1;
  std::vector<DevCtx> devs;
  int cur = 0;  // index into devs of the entry whose streams are mirrored in the members below
  hipStream_t stream = nullptr;
  // MSM pipeline on the library's own stream: sort (memory-bound) | accumulation (VALU-bound) | bucket
  // reduction (latency-bound) of consecutive MSMs run on three internal streams and overlap.
  hipStream_t head_stream = nullptr, accum_stream = nullptr, tail_stream = nullptr;
  std::recursive_mutex mu;
  bool profiling = false;
  std::string prof_filter;
  std::vector<ProfRec> prof;
  char last_err[256] = {0};
};

Ctx& ctx();
// make entry `idx` current: this THREAD's HIP device + the stream members of Ctx mirror its streams (callers hold the mutex)
int use_device(int idx);
// make `device` the calling thread's HIP device (checked against hipGetDevice every time: HIP's current device is per host thread)
int set_thread_device(int device);
// every extern "C" entry point: the calling thread's HIP device = the primary device (entry 0), whatever another thread
// or an earlier call on this thread selected; entry points that drive other devices switch with use_device under the mutex
int bind_thread();
// the primary device's stream, for entry points that do not take the mutex (another thread may be switching the mirrors
// in Ctx between devices inside a sharded MSM)
inline hipStream_t primary_stream() { return ctx().devs.empty() ? nullptr : ctx().devs[0].stream; }

inline hipStream_t pick_stream(h2mi_stream_t s) { return s ? reinterpret_cast<hipStream_t>(s) : ctx().stream; }

void note_hip_error(hipError_t e, const char* file, int line);

#define H2_HIP(x)                                   \
  do {                                              \
    hipError_t e_ = (x);                            \
    if (e_ != hipSuccess) {                         \
      ::h2::note_hip_error(e_, __FILE__, __LINE__); \
      return H2MI_EHIP;                             \
    }                                               \
  } while (0)

// a HIP call on a release / teardown path whose failure cannot change what the caller does next: the status is still
// looked at — recorded for h2mi_strerror and printed under H2MI_VERBOSE — instead of being dropped on the floor
#define H2_IGNORE(x)                                                      \
  do {                                                                    \
    hipError_t ei_ = (x);                                                 \
    if (ei_ != hipSuccess) ::h2::note_hip_error(ei_, __FILE__, __LINE__); \
  } while (0)

#define H2_REQUIRE_INIT()                        \
  do {                                           \
    if (!::h2::ctx().inited) return H2MI_ENODEV; \
    int rcb_ = ::h2::bind_thread();              \
    if (rcb_) return rcb_;                       \
  } while (0)

// Event-bracketed launch: when profiling is on, record a HIP event before and after the kernel on the
// stream it is launched on (device time of exactly this launch); otherwise a plain launch.
void prof_begin(const char* name, hipStream_t s);
void prof_end(hipStream_t s);
inline bool prof_on(const char* name) {
  Ctx& c = ctx();
  return c.profiling && (c.prof_filter.empty() || strncmp(name, c.prof_filter.c_str(), c.prof_filter.size()) == 0);
}

#define H2_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                       \
  do {                                                                                 \
    const bool prof_ = ::h2::prof_on(name);                                            \
    if (prof_) ::h2::prof_begin(name, stream);                                         \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, stream, __VA_ARGS__);   \
    if (prof_) ::h2::prof_end(stream);                                                 \
    H2_HIP(hipGetLastError());                                                         \
  } while (0)

// device allocation released on every exit path of the synchronous helper entry points
struct DevMem {
  void* p = nullptr;
  ~DevMem() { if (p) H2_IGNORE(hipFree(p)); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// A device table built by kernels on one stream and read by kernels on any stream: the builder records an
// event, a consumer on another stream waits for it (same stream: in order, no wait needed).
struct Built {
  hipEvent_t ev = nullptr;
  hipStream_t on = nullptr;
  hipError_t mark(hipStream_t s) {
    if (!ev) {
      hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e != hipSuccess) return e;
    }
    on = s;
    return hipEventRecord(ev, s);
  }
  hipError_t use(hipStream_t s) const { return (ev && s != on) ? hipStreamWaitEvent(s, ev, 0) : hipSuccess; }
  void destroy() {
    if (ev) H2_IGNORE(hipEventDestroy(ev));
    ev = nullptr;
  }
};

// h2mi_core.hip: out[j] = sum over r < world of pts[r*k + j] (Jacobian, Montgomery-2^256), one launch on `s`
int launch_fold_groups(const uint8_t* pts, size_t world, size_t k, uint8_t* out, hipStream_t s);

// h2mi_msm.hip: make `s` wait for all outstanding MSM tails
int msm_join_all(hipStream_t s);
int msm_flush_all();
// per-module teardown hooks of h2mi_shutdown (device memory, events and cached tables of the devices being released)
void msm_teardown();
void ntt_teardown();
void lookup_teardown();

// Tuning / A-B knobs (window overrides, occupancy experiments, measured losers kept for re-measurement) exist only in a library
// built with -DH2MI_AB (`make ab` -> libh2mi_ab.so, used by tools/*sweep* and tools/ab_*.sh): the shipped libh2mi.so holds the
// measured winners and reads no tuning variable.  Configuration that tests and rehearsals rely on (H2MI_VERBOSE,
// H2MI_VIRTUAL_DEVICES, H2MI_MSM_C / H2MI_MSM_S0 / H2MI_MSM_NO_PIPELINE forced-path parity, H2MI_POWTAB_MAX) stays a plain getenv.
#ifdef H2MI_AB
inline const char* ab_env(const char* name) { return getenv(name); }
#else
inline const char* ab_env(const char*) { return nullptr; }
#endif

inline uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

}  // namespace h2
