// libh2mi.so — lifecycle, device memory, profiling and the small G1 helpers.
#include <algorithm>

#include "g1.cuh"
#include <cstdio>

#include "h2mi_internal.h"

namespace h2 {

Ctx& ctx() {
  static Ctx c;
  return c;
}

// HIP's current device is a property of the HOST THREAD, and other code on the same thread may change it between two calls
// (torch.cuda.set_device / a device guard in the embedding process, the Rust shim's host, any other HIP user): every entry point
// asks the runtime what the thread's device IS — hipGetDevice is a thread-local read, no driver call — instead of trusting a
// private cache of what this library last selected (round-4 ADVICE: such a cache skipped the hipSetDevice and sent allocations
// and launches to the embedding code's device).
int set_thread_device(int device) {
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device) H2_HIP(hipSetDevice(device));
  return H2MI_OK;
}

int bind_thread() {
  Ctx& c = ctx();
  if (c.devs.empty()) return H2MI_ENODEV;
  return set_thread_device(c.devs[0].device);
}

int use_device(int idx) {
  Ctx& c = ctx();
  if (idx < 0 || idx >= (int)c.devs.size()) return H2MI_EINVAL;
  const DevCtx& d = c.devs[idx];
  int rc = set_thread_device(d.device);
  if (rc) return rc;
  c.cur = idx;
  c.device = d.device;
  c.stream = d.stream;
  c.head_stream = d.head_stream;
  c.accum_stream = d.accum_stream;
  c.tail_stream = d.tail_stream;
  return H2MI_OK;
}

void note_hip_error(hipError_t e, const char* file, int line) {
  snprintf(ctx().last_err, sizeof(ctx().last_err), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
  if (getenv("H2MI_VERBOSE")) fprintf(stderr, "[h2mi] %s\n", ctx().last_err);
}

void prof_begin(const char* name, hipStream_t s) {
  ProfRec r;
  r.name = name;
  H2_IGNORE(hipEventCreate(&r.a));
  H2_IGNORE(hipEventCreate(&r.b));
  H2_IGNORE(hipEventRecord(r.a, s));
  ctx().prof.push_back(r);
}
void prof_end(hipStream_t s) { H2_IGNORE(hipEventRecord(ctx().prof.back().b, s)); }

// sum of k Jacobian points by one thread (k is tiny: the number of GPUs)
__global__ void k_g1_sum_jac(const uint8_t* pts, size_t k, uint8_t* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  xyzz acc = xyzz_identity();
  for (size_t i = 0; i < k; i++) {
    xyzz p = jac_to_xyzz(jac_load(pts + i * 96));
    xyzz_add(acc, p);
  }
  jac_store(out, xyzz_to_jac(acc));
}

// out[j] = sum over r < world of pts[r*k + j]  (rank-major partial results of k MSMs)
__global__ void __launch_bounds__(64) k_g1_fold_groups(const uint8_t* pts, size_t world, size_t k, uint8_t* out) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  xyzz acc = xyzz_identity();
  for (size_t r = 0; r < world; r++) {
    xyzz p = jac_to_xyzz(jac_load(pts + (r * k + j) * 96));
    xyzz_add(acc, p);
  }
  jac_store(out + j * 96, xyzz_to_jac(acc));
}

__global__ void k_g1_normalize(const uint8_t* pts, size_t k, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  xyzz p = jac_to_xyzz(jac_load(pts + i * 96));
  affine_store(out + i * 64, xyzz_to_affine(p));
}

// ---- fixed-base table of the generator: T[w][d] = d * 2^(8w) * G, d in 1..255 (d = 0 unused) -------
__global__ void __launch_bounds__(256) k_fixed_base_table(uint8_t* table) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // t = w*256 + d
  if (t >= 32 * 256) return;
  uint32_t w = t >> 8, d = t & 255;
  affine g;
  g.x = fe_one<Fq>();
  g.y = fe_dbl<Fq>(fe_one<Fq>());  // (1, 2) in Montgomery form
  affine outp;
  if (d == 0) {
    outp.x = fe_zero();
    outp.y = fe_zero();
  } else {
    // scalar = d << (8w): left-to-right double-and-add over the 8 bits of d, then 8w doublings
    xyzz acc = xyzz_identity();
    for (int b = 7; b >= 0; b--) {
      acc = xyzz_dbl(acc);
      if ((d >> b) & 1u) xyzz_madd(acc, g);
    }
    for (uint32_t i = 0; i < 8 * w; i++) acc = xyzz_dbl(acc);
    outp = xyzz_to_affine(acc);
  }
  affine_store(table + (size_t)t * 64, outp);
}

__global__ void __launch_bounds__(256) k_fixed_base_mul(const fe* scalars, size_t n, const uint8_t* table, uint8_t* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe s = fe_from_mont<FrP>(fe_load(&scalars[i]));
  xyzz acc = xyzz_identity();
  for (int w = 0; w < 32; w++) {
    uint32_t d = (s.v[w >> 2] >> ((w & 3) * 8)) & 255u;
    if (d) {
      affine p = affine_load(table + ((size_t)(w * 256 + d)) * 64);
      xyzz_madd(acc, p);
    }
  }
  affine_store(out + i * 64, xyzz_to_affine(acc));
}

int launch_fold_groups(const uint8_t* pts, size_t world, size_t k, uint8_t* out, hipStream_t s) {
  H2_LAUNCH("k_g1_fold_groups", k_g1_fold_groups, ceil_div_u32(k, 64), 64, 0, s, pts, world, k, out);
  return H2MI_OK;
}

static uint8_t* g_fixed_table = nullptr;
static Built g_fixed_built;

}  // namespace h2

using namespace h2;

extern "C" {

static int create_dev(int device, DevCtx* d) {
  int rc = set_thread_device(device);
  if (rc) return rc;
  d->device = device;
  H2_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
  H2_HIP(hipStreamCreateWithFlags(&d->tail_stream, hipStreamNonBlocking));
  H2_HIP(hipStreamCreateWithFlags(&d->head_stream, hipStreamNonBlocking));
  H2_HIP(hipStreamCreateWithFlags(&d->accum_stream, hipStreamNonBlocking));
  return H2MI_OK;
}

int h2mi_init(int device) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (ctx().inited) return (ctx().devs.size() == 1 && ctx().devs[0].device == device) ? H2MI_OK : H2MI_EINVAL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return H2MI_ENODEV;
  if (device < 0 || device >= count) return H2MI_EINVAL;
  DevCtx d;
  int rc = create_dev(device, &d);
  if (rc) return rc;
  ctx().devs.assign(1, d);
  ctx().cur = -1;
  ctx().inited = true;
  return use_device(0);
}

int h2mi_init_devices(int n_devices) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (n_devices < 1 || n_devices > 16) return H2MI_EINVAL;
  if (ctx().inited) return (int)ctx().devs.size() == n_devices ? H2MI_OK : H2MI_EINVAL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return H2MI_ENODEV;
  const bool virt = getenv("H2MI_VIRTUAL_DEVICES") != nullptr;  // rehearsal: entry i runs on GPU i % count
  if (n_devices > count && !virt) return H2MI_ENODEV;
  std::vector<DevCtx> devs((size_t)n_devices);
  for (int i = 0; i < n_devices; i++) {
    int rc = create_dev(i % count, &devs[(size_t)i]);
    if (rc) return rc;
  }
  // peer access between the primary device and the others (slices of device-resident scalars and the 96-byte
  // partial results travel over xGMI); ignore "already enabled"
  for (int i = 1; i < n_devices; i++) {
    if (devs[(size_t)i].device == devs[0].device) continue;
    int can = 0;
    hipError_t pe = hipDeviceCanAccessPeer(&can, devs[0].device, devs[(size_t)i].device);
    if (pe == hipSuccess && can) {
      set_thread_device(devs[0].device);
      pe = hipDeviceEnablePeerAccess(devs[(size_t)i].device, 0);
      if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) {
        set_thread_device(devs[(size_t)i].device);
        pe = hipDeviceEnablePeerAccess(devs[0].device, 0);
      }
    }
    // without peer access hipMemcpyPeerAsync stages through the host: correct but slow, so say so instead of hiding it
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
      std::fprintf(stderr, "h2mi: no peer access between devices %d and %d (%s): slices and partial results will be staged through the host\n",
                   devs[0].device, devs[(size_t)i].device, hipGetErrorString(pe));
    else if (!can)
      std::fprintf(stderr, "h2mi: devices %d and %d cannot access each other's memory: copies will be staged through the host\n", devs[0].device,
                   devs[(size_t)i].device);
  }
  (void)hipGetLastError();
  ctx().devs = devs;
  ctx().cur = -1;
  ctx().inited = true;
  return use_device(0);
}

int h2mi_device_count(void) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  return ctx().inited ? (int)ctx().devs.size() : 0;
}

static hipEvent_t g_wait_ring[32] = {};  // h2mi_stream_wait's events; destroyed by h2mi_shutdown (they belong to its device)
static unsigned g_wait_next = 0;

void h2mi_shutdown(void) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (!ctx().inited) return;
  use_device(0);
  msm_join_all(ctx().stream);
  H2_IGNORE(hipStreamSynchronize(ctx().stream));
  for (DevCtx& d : ctx().devs) {
    set_thread_device(d.device);
    (void)hipDeviceSynchronize();
  }
  msm_teardown();  // registrations on every device
  use_device(0);
  ntt_teardown();     // plans, power tables, scratch vectors (primary device)
  lookup_teardown();  // counting-sort scratch and its event
  if (g_fixed_table) { H2_IGNORE(hipFree(g_fixed_table)); g_fixed_table = nullptr; g_fixed_built.destroy(); }
  for (hipEvent_t& ev : g_wait_ring)
    if (ev) { H2_IGNORE(hipEventDestroy(ev)); ev = nullptr; }
  for (DevCtx& d : ctx().devs) {
    set_thread_device(d.device);
    (void)hipDeviceSynchronize();
    H2_IGNORE(hipStreamDestroy(d.head_stream));
    H2_IGNORE(hipStreamDestroy(d.accum_stream));
    H2_IGNORE(hipStreamDestroy(d.tail_stream));
    H2_IGNORE(hipStreamDestroy(d.stream));
  }
  ctx().devs.clear();
  ctx().head_stream = ctx().accum_stream = ctx().tail_stream = ctx().stream = nullptr;
  ctx().inited = false;
}

const char* h2mi_strerror(int code) {
  switch (code) {
    case H2MI_OK: return "ok";
    case H2MI_EINVAL: return "invalid argument";
    case H2MI_ENODEV: return "no usable GPU (h2mi_init not called or failed); there is no CPU fallback";
    case H2MI_ENOMEM: return "out of memory";
    case H2MI_EHIP: return ctx().last_err[0] ? ctx().last_err : "HIP runtime error";
    case H2MI_EHANDLE: return "unknown bases handle";
    case H2MI_ERANGE: return "size out of range";
    case H2MI_EUNSAT: return "constraint system not satisfied";
    default: return "unknown error";
  }
}

const char* h2mi_version(void) { return "h2mi 0.1 (gfx950)"; }

int h2mi_malloc(size_t bytes, void** d_ptr) {
  H2_REQUIRE_INIT();
  if (!d_ptr) return H2MI_EINVAL;
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
  if (e == hipErrorOutOfMemory) return H2MI_ENOMEM;
  H2_HIP(e);
  return H2MI_OK;
}
int h2mi_free(void* d_ptr) {
  H2_REQUIRE_INIT();
  H2_HIP(hipFree(d_ptr));
  return H2MI_OK;
}
int h2mi_memcpy_h2d(void* d_dst, const void* src, size_t bytes) {
  H2_REQUIRE_INIT();
  H2_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, primary_stream()));
  H2_HIP(hipStreamSynchronize(primary_stream()));
  return H2MI_OK;
}
int h2mi_memcpy_h2d_async(void* d_dst, const void* src, size_t bytes) {
  H2_REQUIRE_INIT();
  // stream-ordered on the library's stream; for pageable `src` the runtime stages the bytes before it returns,
  // so the caller may reuse the buffer at once (small patches: blinding rows, assigned cells)
  H2_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, primary_stream()));
  return H2MI_OK;
}
// up to PATCH_MAX 32-byte cells travel in the kernel's ARGUMENTS: no staging buffer, no copy engine — one launch
constexpr uint32_t PATCH_MAX = 64;
struct PatchArgs {
  fe* dst[PATCH_MAX];
  fe val[PATCH_MAX];
};
__global__ void __launch_bounds__(64) k_patch_cells(const PatchArgs a, uint32_t count) {
  const uint32_t i = threadIdx.x;
  if (i < count) fe_store(a.dst[i], a.val[i]);
}
int h2mi_fr_patch_cells_dev(void* const* d_cells, const uint64_t* values, size_t count, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (count && (!d_cells || !values)) return H2MI_EINVAL;
  for (size_t i = 0; i < count; i++)
    if (!d_cells[i] || ((uintptr_t)d_cells[i] & 15u)) return H2MI_EINVAL;
  hipStream_t s = stream ? (hipStream_t)stream : primary_stream();
  for (size_t i0 = 0; i0 < count; i0 += PATCH_MAX) {
    const uint32_t m = (uint32_t)std::min<size_t>(PATCH_MAX, count - i0);
    PatchArgs a;
    for (uint32_t i = 0; i < m; i++) {
      a.dst[i] = (fe*)d_cells[i0 + i];
      memcpy(a.val[i].v, values + 4 * (i0 + i), 32);
    }
    for (uint32_t i = m; i < PATCH_MAX; i++) {
      a.dst[i] = a.dst[0];
      a.val[i] = a.val[0];
    }
    hipLaunchKernelGGL(k_patch_cells, dim3(1), dim3(64), 0, s, a, m);
    H2_HIP(hipGetLastError());
  }
  return H2MI_OK;
}
int h2mi_memcpy_d2h(void* dst, const void* d_src, size_t bytes) {
  H2_REQUIRE_INIT();
  {
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    int rc = msm_join_all(primary_stream());
    if (rc) return rc;
  }
  // small read-backs (a phase's points, a proof's evaluations: what every transcript join waits for) land in a pinned buffer of the
  // library's: a copy to pageable memory goes through the runtime's own staging and a second wait
  constexpr size_t PINNED_BYTES = 4096;
  static thread_local void* pinned = nullptr;  // per calling thread: two threads may read back at once
  if (bytes <= PINNED_BYTES && !ab_env("H2MI_NO_PINNED_READBACK")) {
    if (!pinned && hipHostMalloc(&pinned, PINNED_BYTES, hipHostMallocPortable) != hipSuccess) pinned = nullptr;
    if (pinned) {
      H2_HIP(hipMemcpyAsync(pinned, d_src, bytes, hipMemcpyDeviceToHost, primary_stream()));
      H2_HIP(hipStreamSynchronize(primary_stream()));  // (polling hipStreamQuery instead: no difference, measured — the runtime's wait spins)
      memcpy(dst, pinned, bytes);
      return H2MI_OK;
    }
  }
  H2_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, primary_stream()));
  H2_HIP(hipStreamSynchronize(primary_stream()));
  return H2MI_OK;
}
int h2mi_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes) {
  H2_REQUIRE_INIT();
  H2_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, primary_stream()));
  return H2MI_OK;
}
int h2mi_memset_zero(void* d_ptr, size_t bytes) {
  H2_REQUIRE_INIT();
  H2_HIP(hipMemsetAsync(d_ptr, 0, bytes, primary_stream()));
  return H2MI_OK;
}
int h2mi_join(void) {
  H2_REQUIRE_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  return msm_join_all(ctx().stream);
}
int h2mi_msm_flush(void) {
  H2_REQUIRE_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  return msm_flush_all();
}
int h2mi_sync(void) {
  H2_REQUIRE_INIT();
  {
    std::lock_guard<std::recursive_mutex> lk(ctx().mu);
    int rc = msm_join_all(ctx().stream);
    if (rc) return rc;
  }
  H2_HIP(hipStreamSynchronize(ctx().stream));
  return H2MI_OK;
}

int h2mi_profile_enable(int on) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  ctx().profiling = on != 0;
  return H2MI_OK;
}
int h2mi_profile_filter(const char* prefix) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  ctx().prof_filter = prefix ? prefix : "";
  return H2MI_OK;
}
int h2mi_profile_reset(void) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (ctx().inited) H2_IGNORE(hipDeviceSynchronize());
  for (auto& r : ctx().prof) {
    H2_IGNORE(hipEventDestroy(r.a));
    H2_IGNORE(hipEventDestroy(r.b));
  }
  ctx().prof.clear();
  return H2MI_OK;
}
int h2mi_profile_query(const char* prefix, double* total_ms, uint64_t* launches) {
  H2_REQUIRE_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  H2_HIP(hipDeviceSynchronize());
  double tot = 0;
  uint64_t cnt = 0;
  size_t pl = prefix ? strlen(prefix) : 0;
  for (auto& r : ctx().prof) {
    if (pl && r.name.compare(0, pl, prefix) != 0) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      tot += ms;
      cnt++;
    }
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  return H2MI_OK;
}

int h2mi_profile_dump(char* buf, size_t cap, size_t* needed_out) {
  H2_REQUIRE_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  H2_HIP(hipDeviceSynchronize());
  std::string out;
  hipEvent_t first = ctx().prof.empty() ? nullptr : ctx().prof.front().a;
  for (auto& r : ctx().prof) {
    float ms = 0, t0 = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    if (hipEventElapsedTime(&t0, first, r.a) != hipSuccess) t0 = -1;  // events of another device
    char line[160];
    snprintf(line, sizeof(line), "%s %.4f %.4f\n", r.name.c_str(), t0, ms);
    out += line;
  }
  if (needed_out) *needed_out = out.size() + 1;
  if (buf && cap) {
    const size_t nn = std::min(cap - 1, out.size());
    memcpy(buf, out.data(), nn);
    buf[nn] = 0;
  }
  return H2MI_OK;
}

int h2mi_g1_sum_jacobian(const uint64_t* points, size_t k, uint64_t out_jacobian[12]) {
  H2_REQUIRE_INIT();
  if (!points || !out_jacobian || k == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  DevMem dp, dout;
  H2_HIP(dp.alloc(k * 96));
  H2_HIP(dout.alloc(96));
  H2_HIP(hipMemcpyAsync(dp.p, points, k * 96, hipMemcpyHostToDevice, s));
  H2_LAUNCH("k_g1_sum_jac", k_g1_sum_jac, 1, 64, 0, s, dp.as<uint8_t>(), k, dout.as<uint8_t>());
  H2_HIP(hipMemcpyAsync(out_jacobian, dout.p, 96, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  return H2MI_OK;
}


int h2mi_g1_fold_groups(const uint64_t* points, size_t world, size_t k, uint64_t* out) {
  H2_REQUIRE_INIT();
  if (!points || !out || world == 0 || k == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  DevMem dp, dout;
  H2_HIP(dp.alloc(world * k * 96));
  H2_HIP(dout.alloc(k * 96));
  H2_HIP(hipMemcpyAsync(dp.p, points, world * k * 96, hipMemcpyHostToDevice, s));
  H2_LAUNCH("k_g1_fold_groups", k_g1_fold_groups, ceil_div_u32(k, 64), 64, 0, s, dp.as<uint8_t>(), world, k, dout.as<uint8_t>());
  H2_HIP(hipMemcpyAsync(out, dout.p, k * 96, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  return H2MI_OK;
}


int h2mi_g1_fold_groups_dev(const void* d_points, size_t world, size_t k, void* d_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_points || !d_out || world == 0 || k == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  H2_LAUNCH("k_g1_fold_groups", k_g1_fold_groups, ceil_div_u32(k, 64), 64, 0, s, (const uint8_t*)d_points, world, k, (uint8_t*)d_out);
  return H2MI_OK;
}

int h2mi_g1_batch_normalize_dev(const void* d_jac, size_t k, void* d_affine_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_jac || !d_affine_out || k == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  H2_LAUNCH("k_g1_normalize", k_g1_normalize, ceil_div_u32(k, 64), 64, 0, s, (const uint8_t*)d_jac, k, (uint8_t*)d_affine_out);
  return H2MI_OK;
}

// ---- side streams: work that does not depend on the next challenge (the coefficient / extended forms of committed
// columns) can run beside the library stream's chain of small kernels instead of queueing behind it ----------------------
int h2mi_stream_create(h2mi_stream_t* stream_out) {
  H2_REQUIRE_INIT();
  if (!stream_out) return H2MI_EINVAL;
  hipStream_t s = nullptr;
  H2_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream_out = (h2mi_stream_t)s;
  return H2MI_OK;
}
int h2mi_stream_destroy(h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!stream) return H2MI_EINVAL;
  H2_HIP(hipStreamSynchronize((hipStream_t)stream));
  H2_HIP(hipStreamDestroy((hipStream_t)stream));
  return H2MI_OK;
}
int h2mi_stream_wait(h2mi_stream_t waiter, h2mi_stream_t signaller) {
  H2_REQUIRE_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t w = pick_stream(waiter), g = pick_stream(signaller);
  if (w == g) return H2MI_OK;
  // a ring of events: re-recording one only matters to waits issued after the re-record, and a wait issued 32 calls ago
  // has long been consumed by its stream's front end
  hipEvent_t& ev = g_wait_ring[g_wait_next++ & 31u];
  if (!ev) H2_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  H2_HIP(hipEventRecord(ev, g));
  H2_HIP(hipStreamWaitEvent(w, ev, 0));
  return H2MI_OK;
}

int h2mi_library_stream(void** stream_out) {
  H2_REQUIRE_INIT();
  if (!stream_out) return H2MI_EINVAL;
  *stream_out = (void*)ctx().stream;
  return H2MI_OK;
}

int h2mi_g1_batch_normalize(const uint64_t* jacp, size_t k, uint64_t* affine_out) {
  H2_REQUIRE_INIT();
  if (!jacp || !affine_out || k == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  DevMem dp, dout;
  H2_HIP(dp.alloc(k * 96));
  H2_HIP(dout.alloc(k * 64));
  H2_HIP(hipMemcpyAsync(dp.p, jacp, k * 96, hipMemcpyHostToDevice, s));
  H2_LAUNCH("k_g1_normalize", k_g1_normalize, ceil_div_u32(k, 256), 256, 0, s, dp.as<uint8_t>(), k, dout.as<uint8_t>());
  H2_HIP(hipMemcpyAsync(affine_out, dout.p, k * 64, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  return H2MI_OK;
}


int h2mi_g1_fixed_base_mul_dev(const void* d_scalars, size_t n, void* d_out_affine, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_scalars || !d_out_affine || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  if (!g_fixed_table) {
    H2_HIP(hipMalloc(&g_fixed_table, 32 * 256 * 64));
    H2_LAUNCH("k_fixed_base_table", k_fixed_base_table, 32, 256, 0, s, g_fixed_table);
    H2_HIP(g_fixed_built.mark(s));
  }
  H2_HIP(g_fixed_built.use(s));
  H2_LAUNCH("k_fixed_base_mul", k_fixed_base_mul, ceil_div_u32(n, 256), 256, 0, s, (const fe*)d_scalars, n,
            (const uint8_t*)g_fixed_table, (uint8_t*)d_out_affine);
  return H2MI_OK;
}

}  // extern "C"
