// libh2mi.so — the lookup argument's permuted columns on the device (BASELINE config 3).
//
// halo2_proofs plonk/lookup/prover.rs `permute_expression_pair` (reached from create_proof's `lookups.commit_permuted`;
// the reference selects a lookup table with LOOKUP_BITS, src/scaffold.rs:44-48,462): A' = the usable input rows sorted,
// S' = the table rearranged so that S'[j] = A'[j] wherever A' starts a new run and the table values not consumed that
// way fill the repeated rows.  The crate sorts 2^k field elements and walks a BTreeMap on one thread.
//
// Here (single-expression lookups: one input column against one fixed column, the range-check case): every valid input
// value IS a table value, and the table is fixed, so its distinct values are sorted once at keygen (host) and a proof
// needs no sort at all — a counting sort against that table:
//   k_lk_rank         rank of every input in the sorted table (binary search on canonical 256-bit values), histogram
//   scan              run starts of A'
//   k_lk_leftover     table multiplicity minus one for every value that occurs among the inputs; scan -> positions
//   k_lk_fill_input   A'[j] = value of the run containing j; marks repeated rows; scan -> repeated rows before j
//   k_lk_fill_table   S'[j] = A'[j] on a run start, else the (number of repeated rows after j)-th leftover value
//                     (the crate pops repeated rows from the END while walking the leftovers in ascending order)
// Integer / byte work over HBM-resident vectors: no field multiplication except the Montgomery -> canonical conversion
// of the inputs.
#include <algorithm>

#include "fp.cuh"
#include "h2mi_internal.h"
#include "scan.cuh"

namespace h2 {

using Fr = FrP;

__device__ __forceinline__ int cmp256(const fe& a, const fe& b) {  // canonical little-endian words
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    if (a.v[i] != b.v[i]) return a.v[i] < b.v[i] ? -1 : 1;
  }
  return 0;
}

// The histogram is built for the input a range check produces: a few limbs and then millions of zero rows, i.e. nearly
// every thread increments the same counter.  A device-scope atomic on one address retires at ~90 M/s here (measured:
// 47 ms for 2^22 rows), so equal ranks are first merged within the wavefront (ballot; one iteration when the wave is
// uniform), then within the workgroup through a 64-slot LDS table keyed by rank (a slot taken by another rank falls
// back to the global atomic), and only the table is flushed to HBM: 2^22 zero rows -> 4096 global atomics.
constexpr uint32_t LK_SLOTS = 64, LK_EMPTY = 0xFFFFFFFFu;
__global__ void __launch_bounds__(1024) k_lk_rank(const fe* input, uint32_t u, const fe* sorted, uint32_t n_unique, uint32_t* cnt, uint32_t* missing) {
  __shared__ uint32_t hkey[LK_SLOTS], hcnt[LK_SLOTS];
  if (threadIdx.x < LK_SLOTS) {
    hkey[threadIdx.x] = LK_EMPTY;
    hcnt[threadIdx.x] = 0;
  }
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t lo = LK_EMPTY;
  bool active = false;
  if (i < u) {
    const fe v = fe_from_mont<Fr>(fe_load(&input[i]));
    // first index with sorted[idx] >= v.  A range-check table holds 0 .. 2^bits - 1, where a value IS its rank: the guess "rank = low
    // word of v" is tried first (one load instead of seventeen dependent ones); any other table falls through to the binary search
    const uint32_t guess = min(v.v[0], n_unique - 1);
    if (cmp256(fe_load(&sorted[guess]), v) == 0) {
      lo = guess;
    } else {
      uint32_t hi = n_unique;
      lo = 0;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cmp256(fe_load(&sorted[mid]), v) < 0) lo = mid + 1;
        else hi = mid;
      }
    }
    if (lo < n_unique && cmp256(fe_load(&sorted[lo]), v) == 0) {
      active = true;
    } else {
      atomicAdd(missing, 1u);
    }
  }
  const uint32_t lane = threadIdx.x & 63u;
  for (;;) {
    const unsigned long long m = __ballot(active);
    if (!m) break;
    const int leader = __ffsll(m) - 1;
    const uint32_t v = (uint32_t)__shfl((int)lo, leader);
    const unsigned long long same = __ballot(active && lo == v);
    if ((int)lane == leader) {
      const uint32_t c = (uint32_t)__popcll(same);
      const uint32_t slot = v & (LK_SLOTS - 1);
      const uint32_t prev = atomicCAS(&hkey[slot], LK_EMPTY, v);
      if (prev == LK_EMPTY || prev == v) atomicAdd(&hcnt[slot], c);
      else atomicAdd(&cnt[v], c);
    }
    if (active && lo == v) active = false;
  }
  __syncthreads();
  if (threadIdx.x < LK_SLOTS && hkey[threadIdx.x] != LK_EMPTY) atomicAdd(&cnt[hkey[threadIdx.x]], hcnt[threadIdx.x]);
}

__global__ void __launch_bounds__(256) k_lk_leftover(const uint32_t* cnt, const uint32_t* mult, uint32_t n_unique, uint32_t* left, uint32_t* missing) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_unique) return;
  const uint32_t used = cnt[r] ? 1u : 0u;
  if (used > mult[r]) {  // cannot happen for values found in the table; keeps the arithmetic unsigned-safe
    atomicAdd(missing, 1u);
    left[r] = 0;
  } else {
    left[r] = mult[r] - used;
  }
}

// largest r < m with start[r] <= x, where start is an exclusive scan (non-decreasing) and start[m] = total > x
__device__ __forceinline__ uint32_t run_of(const uint32_t* start, uint32_t m, uint32_t x) {
  uint32_t lo = 0, hi = m;  // first index in [0, m] with start[idx] > x
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (start[mid] <= x) lo = mid + 1;
    else hi = mid;
  }
  return lo - 1;
}

__global__ void __launch_bounds__(256) k_lk_fill_input(const uint32_t* start, uint32_t n_unique, const fe* sorted_mont, uint32_t u, fe* a_perm,
                                                       uint32_t* repeated) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= u) return;
  const uint32_t r = run_of(start, n_unique, j);
  fe_store(&a_perm[j], fe_load(&sorted_mont[r]));
  repeated[j] = j != start[r] ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_lk_fill_table(const uint32_t* repeated, const uint32_t* rep_before, const uint32_t* lstart, uint32_t n_unique,
                                                       const fe* sorted_mont, const fe* a_perm, uint32_t u, fe* s_perm) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= u) return;
  if (!repeated[j]) {
    fe_store(&s_perm[j], fe_load(&a_perm[j]));
    return;
  }
  const uint32_t after = rep_before[u] - rep_before[j] - 1;  // repeated rows with a larger index
  fe_store(&s_perm[j], fe_load(&sorted_mont[run_of(lstart, n_unique, after)]));
}

static uint32_t* g_lk_scratch = nullptr;
static size_t g_lk_words = 0;
static hipEvent_t g_lk_event = nullptr;     // last use of the scratch: a caller on another stream queues behind it
static hipStream_t g_lk_stream = nullptr;

static int scan_u32(const uint32_t* in, uint32_t* out, uint32_t m /* multiple of 4 */, uint32_t* segsum, hipStream_t s) {
  const uint32_t nseg = ceil_div_u32(m, SCAN_SEG_BINS);
  if (nseg > 1) H2_LAUNCH("k_scan_segsum", k_scan_segsum<SCAN_SEG_BINS>, nseg, 1024, 0, s, in, m, segsum);
  H2_LAUNCH("k_scan_seg_lookup", k_scan_seg<SCAN_SEG_BINS>, dim3(nseg, 1), 1024, 0, s, in, out, (const uint32_t*)nullptr, (uint32_t*)nullptr, m, (const uint32_t*)segsum);
  return H2MI_OK;
}

// h2mi_shutdown: the scratch and its event belong to the device that is being torn down
void lookup_teardown() {
  if (g_lk_scratch) H2_IGNORE(hipFree(g_lk_scratch));
  if (g_lk_event) H2_IGNORE(hipEventDestroy(g_lk_event));
  g_lk_scratch = nullptr;
  g_lk_words = 0;
  g_lk_event = nullptr;
  g_lk_stream = nullptr;
}

}  // namespace h2

using namespace h2;

extern "C" {

int h2mi_plonk_lookup_permute_dev(const void* d_input, const void* d_table_sorted, const void* d_table_sorted_mont, const void* d_table_mult,
                                  uint32_t n_unique, uint32_t k, uint32_t usable_rows, void* d_permuted_input, void* d_permuted_table,
                                  uint64_t* not_in_table_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  // not_in_table_out is mandatory: with inputs outside the table the permuted columns are not a permutation at all (the
  // crate fails the proof with ConstraintSystemFailure), so a caller must not be able to overlook the count
  if (!d_input || !d_table_sorted || !d_table_sorted_mont || !d_table_mult || !d_permuted_input || !d_permuted_table || !not_in_table_out ||
      n_unique == 0)
    return H2MI_EINVAL;
  if (k == 0 || k > H2MI_MAX_LOG_N || usable_rows == 0 || usable_rows >= ((uint64_t)1 << k) || n_unique > usable_rows) return H2MI_ERANGE;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  {
    int rc0 = use_device(0);  // single-process n-device mode: the argument's vectors live on the primary device
    if (rc0) return rc0;
  }
  hipStream_t s = pick_stream(stream);
  const uint32_t u = usable_rows;
  const uint32_t mu = (n_unique + 3u) & ~3u, uu = (u + 3u) & ~3u;  // scan lengths (multiples of 4; the pads are zero)
  const uint32_t nseg = ceil_div_u32(std::max(mu, uu), SCAN_SEG_BINS) + 1;
  // scratch (words): cnt[mu+4] start[mu+4] left[mu+4] lstart[mu+4] rep[uu+4] rep_before[uu+4] segsum[nseg] missing[4]
  const size_t words = 4 * ((size_t)mu + 4) + 2 * ((size_t)uu + 4) + nseg + 4;
  if (g_lk_words < words) {
    if (g_lk_scratch) {
      H2_HIP(hipDeviceSynchronize());
      H2_HIP(hipFree(g_lk_scratch));
      g_lk_scratch = nullptr;
      g_lk_words = 0;
    }
    hipError_t e = hipMalloc((void**)&g_lk_scratch, words * 4);
    if (e == hipErrorOutOfMemory) return H2MI_ENOMEM;
    H2_HIP(e);
    g_lk_words = words;
  }
  uint32_t* cnt = g_lk_scratch;
  uint32_t* start = cnt + mu + 4;
  uint32_t* left = start + mu + 4;
  uint32_t* lstart = left + mu + 4;
  uint32_t* rep = lstart + mu + 4;
  uint32_t* rep_before = rep + uu + 4;
  uint32_t* segsum = rep_before + uu + 4;
  uint32_t* missing = segsum + nseg;
  if (g_lk_event && g_lk_stream != s) H2_HIP(hipStreamWaitEvent(s, g_lk_event, 0));
  H2_HIP(hipMemsetAsync(g_lk_scratch, 0, words * 4, s));
  const fe* in = (const fe*)d_input;
  const fe* sorted = (const fe*)d_table_sorted;
  const fe* sorted_mont = (const fe*)d_table_sorted_mont;
  H2_LAUNCH("k_lk_rank", k_lk_rank, ceil_div_u32(u, 1024), 1024, 0, s, in, u, sorted, n_unique, cnt, missing);
  int rc = scan_u32(cnt, start, mu, segsum, s);
  if (rc) return rc;
  H2_LAUNCH("k_lk_leftover", k_lk_leftover, ceil_div_u32(n_unique, 256), 256, 0, s, (const uint32_t*)cnt, (const uint32_t*)d_table_mult, n_unique, left, missing);
  rc = scan_u32(left, lstart, mu, segsum, s);
  if (rc) return rc;
  H2_LAUNCH("k_lk_fill_input", k_lk_fill_input, ceil_div_u32(u, 256), 256, 0, s, (const uint32_t*)start, n_unique, sorted_mont, u, (fe*)d_permuted_input, rep);
  rc = scan_u32(rep, rep_before, uu, segsum, s);
  if (rc) return rc;
  // rep_before[uu] holds the total; the fill kernel reads it at index u: identical when u is a multiple of 4, else the
  // zero pads make rep_before[u] == rep_before[uu]
  H2_LAUNCH("k_lk_fill_table", k_lk_fill_table, ceil_div_u32(u, 256), 256, 0, s, (const uint32_t*)rep, (const uint32_t*)rep_before, (const uint32_t*)lstart,
            n_unique, sorted_mont, (const fe*)d_permuted_input, u, (fe*)d_permuted_table);
  if (!g_lk_event) H2_HIP(hipEventCreateWithFlags(&g_lk_event, hipEventDisableTiming));
  H2_HIP(hipEventRecord(g_lk_event, s));
  g_lk_stream = s;
  uint32_t m = 0;  // the crate fails the proof (ConstraintSystemFailure) when an input is not in the table
  H2_HIP(hipMemcpyAsync(&m, missing, 4, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  *not_in_table_out = m;
  return H2MI_OK;
}

}  // extern "C"
