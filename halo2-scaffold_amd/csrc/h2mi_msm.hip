// libh2mi.so — multi-scalar multiplication over BN254 G1 on gfx950 (Pippenger bucket method).
//
// Replaces halo2_proofs::arithmetic::best_multiexp for C = bn256::G1Affine as reached through
// ParamsKZG::commit / commit_lagrange inside create_proof and keygen_vk (SURVEY.md 8a rows a2/a5;
// reference call sites examples/standard_plonk.rs:33,41-49 and src/scaffold.rs:132,191-199,322-331).
//
// MI355X-first design (not the reference's per-thread serial Pippenger):
//  * bases are registered once and kept in HBM together with the table 2^(c*w) * P_i for every window
//    w (W x n x 64 B; 0.94 GiB at n = 2^20, c = 17 — HBM is 288 GB).  All windows then feed ONE set of
//    2^(c-1) buckets, so there is no per-window reduction and no c-doubling Horner chain;
//  * scalars: Montgomery -> canonical, signed c-bit digits (halves the bucket count), zero digits skipped;
//  * (bucket, point) pairs are grouped by bucket with a two-level counting partition in LDS (tile ->
//    512 bins -> buckets; see "bucket partition" below) instead of a general key/value radix sort; global
//    atomics are avoided: on gfx950 a device-scope atomic is a 64-B memory-side transaction (~24 G/s
//    measured), far slower than staging through LDS;
//  * bucket accumulation is load-balanced: the bucket-ordered entry array is cut into equal chunks of <= 128
//    entries across bucket boundaries, one thread per chunk, mixed additions in XYZZ coordinates on 64-B
//    gathered table points, so a hot bucket (witness columns full of 0/1) never serialises a wavefront and
//    every lane does the same work; partial sums are folded by fixed-depth segmented levels;
//  * the weighted bucket sum  sum_b (b+1) B_b  is done without long serial chains: row/column sums of
//    the bucket matrix (LDS tree reductions), bit-decomposed weights, then <= 15 doublings.
// All arithmetic is 254-bit integer work on v_mad_u64_u32; no MFMA (not a dense contraction).

#include <algorithm>
#include <map>
#include <vector>

#include "g1.cuh"
#include "g1_29.cuh"
#include "g1_29_quad.cuh"
#include "h2mi_internal.h"
#include "scan.cuh"

namespace h2 {

constexpr uint32_t S0_MAX = 128;  // most entries per accumulation chunk (see accum_chunk_len)
constexpr uint32_t S1 = 8;       // partials per fold task
constexpr uint32_t FG_NARROW = 8;  // workers (lanes or quads) that cooperate on one bucket in k_msm_finish
constexpr uint32_t FG_WIDE = 2;    // wide windows (2^17 .. 2^19 buckets holding a partial or two each): two lanes per bucket
constexpr size_t SHIFT_MIN_N = 4096;  // base sets below this never use the dominant-value shift
constexpr uint32_t HOT_MIN = 64; // a bucket with more folded partials than this gets a whole workgroup (k_msm_finish_hot)
constexpr uint32_t HOT_MIN_WIDE = 8;  // wide windows: two lanes per bucket, no fold level (the finish reads the accumulation's partials)

// Entries per accumulation chunk (= additions per thread).  The accumulation is resident at two workgroups of
// 256 threads per CU (the occupancy cap in msm_dev): ACCUM_RESIDENT_CHUNKS chunks run at once, all of the same
// length, so the kernel's time is (rounds of resident chunks) x (chunk length).  The length is chosen so that the
// chunks fill a whole number of rounds — with a power-of-two length a 2^18 / 2^19 slice ran 1.06 rounds, a
// second, nearly empty round of full-length chains (0.45 / 0.88 ms instead of 0.36 / 0.67 ms) — and from the
// number of entries the partition actually produced (known on the device only): the advice columns of a real
// witness are mostly zero, and a length derived from n * W would leave a tenth of the threads with 128-entry
// chains.  Up to 128 entries: at 2^20 one round of 128 costs what two rounds of 64 did and leaves half as many
// partial sums to fold.
constexpr uint32_t ACCUM_RESIDENT_CHUNKS = 256 * 2 * 256;
__host__ __device__ inline uint32_t accum_rounds(uint32_t entries) {
  const uint32_t per_round = S0_MAX * ACCUM_RESIDENT_CHUNKS;  // 2^24
  return entries ? (entries + per_round - 1) / per_round : 1;
}
// Nearly empty columns (the advice columns of a padded circuit: a few dozen entries): one entry per chunk.  The
// accumulation then only gathers (an accumulator that starts at the identity copies its first point) and the additions
// happen in the fold / finish kernels, whose lane-cooperative operations are built for short dependent chains.  With
// eight entries per chunk a handful of lone wavefronts walked the accumulation's 35 KB loop body through a cold
// instruction cache: 0.6 ms for 144 entries (measured), whatever the entry count.
constexpr uint32_t ACCUM_GATHER_ONLY_MAX = 16384;
__host__ __device__ inline uint32_t accum_chunk_len(uint32_t entries) {
  if (entries <= ACCUM_GATHER_ONLY_MAX) return 1;
  const uint32_t slots = accum_rounds(entries) * ACCUM_RESIDENT_CHUNKS;
  const uint32_t s0 = (entries + slots - 1) / slots;
  return s0 < 8 ? 8 : s0;
}

// per-call workspace; several slots per handle so that consecutive MSMs overlap (partition of one beside
// the accumulation of another) and their bucket reductions can be deferred and run as one batch.
struct Slot {
  uint32_t* vals[2] = {nullptr, nullptr};     // sign<<31 | w*n_reg+i : [0] grouped by bin, [1] by bucket
  uint8_t* bkeys = nullptr;     // partition intermediate: bucket id within the bin, bin-major
  uint32_t* bincnt = nullptr;   // [nbins][ntiles] entries per (bin, tile), + 1 trailing zero
  uint32_t* binbase = nullptr;  // its exclusive scan; [nbins * ntiles] = number of entries
  uint32_t* binseg = nullptr;   // sums of the SCAN_SEG_BINS-cell segments of bincnt
  uint32_t* tile_live = nullptr;  // per partition tile: does it hold a non-zero scalar (k_msm_bin_count -> k_msm_bin_scatter)
  uint32_t* off = nullptr;      // first entry of each bucket in vals[1] (nb+1; last = number of entries)
  uint32_t* s0_dev = nullptr;   // chunk length of this MSM's accumulation (chosen by k_msm_bin_sort)
  uint32_t* np[2] = {nullptr, nullptr};    // per bucket: partial sums the accumulation leaves, fold tasks (nb+1 entries, last = 0)
  uint32_t* toff[2] = {nullptr, nullptr};  // exclusive scans of np (nb+1 entries, last = total)
  uint8_t* part[2] = {nullptr, nullptr};   // XYZZ partial buffers: accumulation output, fold output
  uint8_t* dense = nullptr;                // one XYZZ sum per bucket
  uint8_t *dense2 = nullptr, *vsum = nullptr;  // wide windows: per segment of 2^seg_log buckets, plain and local weighted sums
  uint32_t* tseg[2] = {nullptr, nullptr};      // wide windows: segment sums of the two task-count scans
  uint8_t* rc = nullptr;                            // row sums [Nh] then column sums [Nl]
  uint8_t* g = nullptr;                             // weighted partials (<= 32)
  uint64_t* stats = nullptr;                        // [0] = insertions
  fe* shift = nullptr;                              // the dominant scalar value this MSM subtracts (k_msm_pick_shift), Montgomery
  hipEvent_t input_ready = nullptr, head_done = nullptr, accum_done = nullptr, tail_done = nullptr;
  bool tail_pending = false, accum_pending = false, head_pending = false;
  bool tail_ever = false, accum_ever = false, head_ever = false;  // the events have been recorded at least once
  hipStream_t last_stream = nullptr;  // stream the slot's previous MSM was issued on
  bool tail_deferred = false;   // accumulation queued, bucket reduction not yet launched (see flush_tails)
  void* d_out = nullptr;        // where that reduction will write the result
  uint32_t tasks1 = 0;          // its fold grid bound
  // small base sets (see "small base sets" below): signed digit bytes [W][n], per-workgroup partial sums and entry counts
  uint8_t* sdig = nullptr;
  uint8_t* spart = nullptr;
  uint32_t* scnt = nullptr;
  bool small_deferred = false;  // the deferred work of this slot is the small path's accumulate + final pair
  uint32_t small_n = 0;         // points of that MSM
  // pair-affine accumulation (H2MI_MSM_PA, used by the -DH2MI_AB library only): products before each pair [PA_MAX_PAIRS][9][pa_T], chunk totals [9][pa_T]
  uint32_t *pa_spill = nullptr, *pa_tot = nullptr;
  uint32_t pa_T = 0;
};
// slots per handle = MSMs that can be in flight between two joins before a flush is forced
// Eight for base sets up to 2^17 (round 3): with four, every fifth back-to-back MSM waited for the reduction batch of the four
// before it — the pipeline drained every four MSMs (2^17: 290 -> 262 us per MSM, 2^16: 214 -> 171, 2^14: 134 -> 105; the
// accumulation alone is 167 us at 2^17).  A prover phase queues at most four commitments, so this matters to commitment streams
// (keygen, many-column circuits, 8-GPU slices), not to the proofs measured here.  From 2^18 four: the 2^20 replay step was 1 %
// SLOWER with eight (19.6 - 19.7 vs 19.8 - 19.9 ms, three alternating pairs on one box: twice the workspace to walk through).
constexpr int NSLOT = 8;

struct Bases {
  int dev = 0;                  // index into ctx().devs of the device that owns every allocation below
  size_t n = 0;
  size_t stride = 0;            // table row length: n + 1 (slot n = the sum of all n bases, see k_msm_pick_shift)
  bool has_sum = false;
  uint32_t c = 0, W = 0, nb = 0, logNl = 0, logNh = 0;
  uint32_t seg_log = 0;         // wide windows (c >= 18): nb = 2^(MAT_LOG + seg_log); the bucket matrix stays 2^logNh x 2^logNl = 2^16
  uint8_t* table = nullptr;     // [W][n] affine, 64 B each (canonical Montgomery-2^261 words)
  uint8_t* host_stage = nullptr;  // 96 B result + n scalars: staging of the host-pointer entry point (lazy)
  uint32_t lb = 0, nbins = 0;   // partition: nbins bins of 2^lb buckets
  Slot slot[NSLOT];
  int nslot = NSLOT;            // slots in use (allocated) for this handle
  int next_slot = 0, last_slot = 0;
  uint32_t max_tasks0 = 0, max_tasks1 = 0;
  // small base sets (small_geometry): second table holding every multiple a signed digit can select, no buckets
  bool small = false;
  uint32_t sc = 0, sW = 0, slanes = 0, sr = 0, sG = 0;  // window bits, windows, gathering lanes, cells per lane, workgroups (= partial sums)
  uint8_t* stable = nullptr;    // [2^(sc-1)][sW][n] affine: j 2^(sc w) P_i
  uint32_t since_join = 0;      // MSMs issued on this handle since the last join (msm_join_all): how deep the caller's queue is
  bool last_small = false;      // the path the last MSM took (h2mi_msm_last_stats)
};

static std::map<uint64_t, Bases*> g_bases;
// cache of ad-hoc registrations (h2mi_msm_bn254_g1 with handle = 0): see adhoc_handle
struct AdHoc {
  size_t n;
  uint64_t fp[2];
  uint64_t handle;
  uint64_t last_use;
};
static std::vector<AdHoc> g_adhoc;
static uint64_t g_adhoc_clock = 0, g_adhoc_builds = 0;
constexpr size_t ADHOC_MAX = 4;  // g, g_lagrange and a couple of slices
static bool g_canonical = false;
// Which pipeline a base set with a digit-multiples table takes, per MSM (round 5; rounds 3-4 chose by size alone).  The latency
// path wins a LONE commitment and a prover phase of up to four at every size it is built for, but from 2^13 points its 2.5x more mixed
// additions lose to the general pipeline once MSMs stream back to back (2^14: 116.7 vs 96.0 us per MSM, profiles/r04_msm_sweep.txt) —
// the workload of an 8-GPU rank's 2^13 .. 2^14-point slices of a 2^16 / 2^17-row proof.  So: the caller can force the general pipeline
// (H2MI_MSM_GENERAL on the phase entry), and without the flag a base set above SMALL_STREAM_N points switches to it after
// SMALL_STREAM_AFTER MSMs have been issued without a join — a phase (<= 4 commitments, then the transcript needs them) never gets
// there, a stream does after its first four.  Results do not depend on the path (tests/test_gpu_parity.py runs both against the oracle).
// Measured (profiles/r05_msm_sweep.txt; latency / four MSMs + join / back to back, us; latency path pinned | general | this rule):
//   2^13  151 / 356 / 85.1 | 266 / 546 / 87.1 | stays on the latency path      2^14  182 / 489 / 113.2 | 271 / 576 / 97.8 | 182 / 483 / 98.5
constexpr size_t SMALL_STREAM_N = (size_t)1 << 13;
constexpr uint32_t SMALL_STREAM_AFTER = 4;
static uint64_t g_next_handle = 1;

// ---- registration: table[w][i] = 2^(c*w) * P_i, stored as canonical Montgomery-2^261 words -------------
// window 0: the caller's points (Montgomery-2^256) converted to the table format
__global__ void __launch_bounds__(256) k_msm_table_first(const uint8_t* bases, uint8_t* table0, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine p = affine_load(bases + i * 64);
  if (!affine_is_identity(p)) {
    affine q;
    f29_pack(f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(p.x.v)), q.x.v);
    f29_pack(f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(p.y.v)), q.y.v);
    p = q;
  }
  affine_store(table0 + i * 64, p);
}
// window w from window w-1: c doublings in XYZZ, one inversion back to affine
__global__ void __launch_bounds__(256) k_msm_table_next(const uint8_t* prev, uint8_t* next, size_t n, uint32_t c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine p = affine_load(prev + i * 64);
  if (affine_is_identity(p)) {
    affine_store(next + i * 64, p);
    return;
  }
  xyzz29 acc = xyzz29_dbl_affine(f29_unpack(p.x.v), f29_unpack(p.y.v));
  for (uint32_t k = 1; k < c; k++) acc = xyzz29_dbl(acc);
  f29 x, y;
  xyzz29_to_affine(acc, x, y);
  affine q;
  f29_pack(x, q.x.v);
  f29_pack(y, q.y.v);
  affine_store(next + i * 64, q);
}

// table slot n of every window: 2^(c*w) * B for B = the sum of the n bases (a Jacobian Montgomery-2^256 point, the
// result of an all-ones MSM at registration); thread w does its c*w doublings and one inversion
__global__ void __launch_bounds__(64) k_msm_table_sum_point(const uint8_t* sum_jac, uint8_t* table, size_t stride, size_t n, uint32_t W, uint32_t c) {
  const uint32_t w = threadIdx.x;
  if (w >= W) return;
  const jac j = jac_load(sum_jac);
  affine out;
  out.x = fe_zero();
  out.y = fe_zero();
  if (!fe_is_zero(j.z)) {
    const affine a = xyzz_to_affine(jac_to_xyzz(j));  // Montgomery-2^256
    f29 x = f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(a.x.v)), y = f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(a.y.v));
    if (w) {
      xyzz29 acc = xyzz29_dbl_affine(x, y);
      for (uint32_t k = 1; k < c * w; k++) acc = xyzz29_dbl(acc);
      xyzz29_to_affine(acc, x, y);
    }
    f29_pack(x, out.x.v);
    f29_pack(y, out.y.v);
  }
  affine_store(table + ((size_t)w * stride + n) * 64, out);
}
__global__ void __launch_bounds__(256) k_msm_fill_one(fe* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fe_store(&out[i], fe_one<FrP>());
}

// ---- per call -------------------------------------------------------------------------------------
// ---- bucket partition (the head of an MSM) -------------------------------------------------------
// Groups the (bucket, point) pairs by bucket with a two-level counting partition instead of a general
// radix sort; zero digits are never materialised.  NBINS top-level bins of 2^lb buckets each:
//   k_msm_bin_count    per tile of P1_TS scalars: entries per bin                 (reads the scalars)
//   exclusive scan     over [bin][tile] -> each tile's write offset in each bin
//   k_msm_bin_scatter  recomputes the digits, groups the tile's entries by bin in LDS and writes each group as
//                      one contiguous run: payload (u32) + bucket id within the bin (u8)
//   k_msm_bin_sort     one workgroup per bin: counts its 2^lb buckets, writes the bucket tables (off,
//                      task counts) and moves the payloads into bucket order, LDS-staged in chunks.
// HBM traffic per MSM at n = 2^20, W = 16: 2 x 32 MiB scalar reads, 84 MB + 84 MB for the intermediate
// pairs, 67 MB of sorted payloads: 0.3 GB against 0.7 GB + 0.1 GB for key/value radix sorting.
// The order of the points inside a bucket follows LDS-atomic arrival, so it is not reproducible: see
// k_msm_final for what that means for the result.
constexpr uint32_t P1_TS = 768;       // scalars per partition tile = threads per workgroup (512, whose 48-KB scatter stage fits beside two
                                      // accumulation workgroups on a CU, measured again in round 4: no difference — profiles/r04_accum_ab.txt)
constexpr uint32_t NBINS_MAX = 512;
// Wide windows (c = 18 .. 20: 2^17 .. 2^19 buckets; round 3, measured in profiles/r03_msm_sweep.txt).  The first partition
// level keeps its 512 bins, so a bin holds up to 1024 buckets: 16-bit in-bin keys and a 1024-entry second-level histogram
// (k_msm_bin_sort_wide).  The bucket-matrix kernels (rowcol / weighted / final) keep their 2^16-bucket matrix: k_msm_seg first
// folds every run of L = 2^(c-17) consecutive buckets into its plain sum P_s and its local weighted sum V_s, and
//     sum_b (b + 1) B_b = L sum_s (s + 1) P_s - sum_s V_s,   V_s = sum_j (L - 1 - j) B_(sL+j).
constexpr uint32_t WIDE_MIN_C = 18, WIDE_MAX_C = 20, MAT_LOG = 16;
// Wide windows bin a bucket by its LOW nine bits (in-bin key = the rest): the top window of a 254-bit scalar holds only
// 254 mod c bits (14 at c = 20, 7 at c = 19, 2 at c = 18), so its digits fall into the lowest buckets — with the narrow
// windows' high-bit binning, into 16 (or one) of the 512 bins, whose workgroups then sort a thirteenth of the MSM alone
// (measured: k_msm_bin_sort 185 us at c = 20, 1.3 - 1.6 ms at c = 18 / 19, against ~55 us balanced).  The sorted array is
// therefore in PERMUTED bucket order b' = (b & 511) << lb | b >> 9; accumulation, fold and finish never look at a bucket's
// number, and k_msm_seg undoes the permutation when it gathers its segments.
constexpr uint32_t WIDE_BIN_BITS = 9;
template <bool WIDE>
__device__ __forceinline__ uint32_t bin_of(uint32_t bucket, uint32_t lb) { return WIDE ? (bucket & ((1u << WIDE_BIN_BITS) - 1)) : (bucket >> lb); }
template <bool WIDE>
__device__ __forceinline__ uint32_t key_of(uint32_t bucket, uint32_t mask) { return WIDE ? (bucket >> WIDE_BIN_BITS) : (bucket & mask); }
constexpr uint32_t P2_THREADS = 512, P2_PER = 16, P2_CH = P2_THREADS * P2_PER;

// Windows: a canonical scalar has 254 bits and the signed recoding can carry one into the top window, so
// W = ceil(255 / c) windows suffice (the top digit then stays <= 2^(c-1): never negative, no carry out).
// CT = compile-time window width (13, 15, 16, 17: fully unrolled, limb indices become constants) or 0 = use
// the run-time c (H2MI_MSM_C experiments only: the dynamically indexed limbs then live in LDS/scratch)
template <uint32_t CT, class F>
__device__ __forceinline__ void for_each_digit(const fe& s, uint32_t c_rt, uint32_t W_rt, F&& f) {
  const uint32_t c = CT ? CT : c_rt;
  const uint32_t W = CT ? (255 + CT - 1) / CT : W_rt;
  uint32_t carry = 0;
  const uint32_t half = 1u << (c - 1);
#pragma unroll
  for (uint32_t w = 0; w < W; w++) {
    uint32_t bit = w * c;
    uint32_t limb = bit >> 5, sh = bit & 31;
    uint32_t raw = 0;
    if (limb < 8) {
      uint64_t two = s.v[limb];
      if (limb + 1 < 8) two |= (uint64_t)s.v[limb + 1] << 32;
      raw = (uint32_t)(two >> sh) & ((1u << c) - 1);
    }
    uint32_t d = raw + carry;  // 0 .. 2^c
    uint32_t mag = d, neg = 0;
    carry = 0;
    if (d > half) {  // negative digit d - 2^c, carry 1
      mag = (1u << c) - d;
      neg = 1;
      carry = 1;
    }
    if (mag) f(w, mag - 1, neg);
  }
}


// Dominant-value shift.  Real prover columns are often one value repeated (a permutation grand product is constant
// wherever a row takes no part in a copy constraint — almost everywhere in a padded circuit — and selector-like
// columns are runs of one constant): every window of such a scalar lands in ONE bucket, and the whole MSM in W
// buckets.  With B = sum_i P_i registered as an extra base (table slot n),
//     sum_i s_i P_i = sum_i (s_i - v) P_i + v B,
// so subtracting the majority value v turns the column into a sparse one (zero digits are never materialised) plus
// one more scalar.  k_msm_pick_shift samples 64 evenly spaced scalars and takes v = the value held by >= 40 of them,
// else 0 (uniform data: nothing changes).  Only for MSMs over all registered bases (B is their sum).
__device__ __forceinline__ void msm_pick_shift_body(const fe* scalars, size_t n, fe* shift_out) {
  __shared__ fe smp[64];
  const uint32_t lane = threadIdx.x;
  size_t idx = (n / 64) * lane + n / 128;
  if (idx >= n) idx = n - 1;
  const fe v = fe_load(&scalars[idx]);
  smp[lane] = v;
  __syncthreads();
  uint32_t same = 0;
  for (uint32_t j = 0; j < 64; j++) same += fe_eq(v, smp[j]) ? 1u : 0u;
  const uint64_t winners = __ballot(same >= 40);
  if (lane == 0) fe_store(shift_out, winners ? smp[__ffsll((unsigned long long)winners) - 1] : fe_zero());
}
__global__ void __launch_bounds__(64) k_msm_pick_shift(const fe* scalars, size_t n, fe* shift_out) { msm_pick_shift_body(scalars, n, shift_out); }
// scalar i of an MSM over n (+ 1) points as canonical integer: s_i - shift for the caller's scalars, shift itself for
// the sum point
// A zero scalar (an unassigned row; every row of a constant column after the shift) skips the Montgomery conversion: real
// columns are mostly zero — the advice, permuted-lookup and (shifted) grand-product columns of the DEGREE 22 range proof spent
// 0.4 + 0.9 ms each in the two digit kernels converting four million zeros (round 3, profiles/r03 timeline) — and a
// wavefront whose 64 scalars are all zero now does no field arithmetic at all.
__device__ __forceinline__ fe msm_scalar(const fe* scalars, size_t i, size_t n, const fe* shift) {
  fe x;
  if (!shift) {
    x = fe_load(&scalars[i]);
  } else {
    const fe v = fe_load(shift);
    x = i < n ? fe_sub<FrP>(fe_load(&scalars[i]), v) : v;
  }
  if (fe_is_zero(x)) return x;
  // Montgomery -> canonical on the 29-bit-limb layer: (x 2^256) * 2^5 / 2^261 = x.  The multiplier has ONE non-zero limb, so the
  // product half is 9 multiply-adds and the whole conversion ~210 instructions against ~400 for fe_from_mont's 32-bit-limb
  // multiplication by one (both digit kernels convert every scalar: -11 us per 2^20 MSM).
  f29 c = f29_zero();
  c.v[0] = 32;
  fe o;
  f29_pack(f29_reduce_canonical<Fr29>(f29_mul<Fr29>(f29_unpack(x.v), c)), o.v);
  return o;
}

template <uint32_t CT>
__device__ __forceinline__ void msm_bin_count_body(const fe* scalars, size_t n, const fe* shift, uint32_t c, uint32_t W, uint32_t lb, uint32_t nbins,
                                                   uint32_t ntiles, uint32_t* cnt_out, uint32_t* tile_live) {
  H2_AB_PRIO();
  __shared__ uint32_t cnt[NBINS_MAX];
  __shared__ uint32_t any_live;
  const uint32_t tid = threadIdx.x;
  if (tid < NBINS_MAX) cnt[tid] = 0;
  if (tid == 0) any_live = 0;
  __syncthreads();
  size_t i = (size_t)blockIdx.x * P1_TS + tid;
  if (i < n + (shift ? 1 : 0)) {
    fe s = msm_scalar(scalars, i, n, shift);
    if (!fe_is_zero(s)) {
      any_live = 1;  // benign race: every writer stores the same value
      for_each_digit<CT>(s, c, W, [&](uint32_t, uint32_t bucket, uint32_t) { atomicAdd(&cnt[bin_of<(CT >= WIDE_MIN_C)>(bucket, lb)], 1u); });
    }
  }
  __syncthreads();
  // a tile of zero scalars (most tiles of a sparse column) has nothing to scatter: k_msm_bin_scatter returns on this flag before
  // it reads a single scalar — the partition of a sparse column is then ONE pass over the scalars, not two
  if (tid == 0) tile_live[blockIdx.x] = any_live;
  if (tid < nbins) cnt_out[(size_t)tid * ntiles + blockIdx.x] = cnt[tid];
  if (blockIdx.x == 0 && tid == 0) cnt_out[(size_t)nbins * ntiles] = 0;  // the scan leaves the total there
}
template <uint32_t CT>
__global__ void __launch_bounds__(P1_TS) k_msm_bin_count(const fe* scalars, size_t n, const fe* shift, uint32_t c, uint32_t W, uint32_t lb,
                                                         uint32_t nbins, uint32_t ntiles, uint32_t* cnt_out, uint32_t* tile_live) {
  msm_bin_count_body<CT>(scalars, n, shift, c, W, lb, nbins, ntiles, cnt_out, tile_live);
}

extern __shared__ uint4 h2_msm_smem[];

template <uint32_t CT>
__device__ __forceinline__ void msm_bin_scatter_body(const fe* scalars, size_t n, const fe* shift, size_t n_reg, uint32_t c, uint32_t W, uint32_t lb,
                                                     uint32_t nbins, uint32_t ntiles, const uint32_t* base, uint32_t* vals_out, void* keys_out_,
                                                     const uint32_t* tile_live) {
  H2_AB_PRIO();
  constexpr bool WIDE = CT >= WIDE_MIN_C;  // in-bin keys of up to 10 bits: staged and written as 16-bit values
  if (!tile_live[blockIdx.x]) return;  // no non-zero scalar in this tile (k_msm_bin_count): block-uniform, before any barrier
  __shared__ uint32_t cnt[NBINS_MAX], lstart[NBINS_MAX + 1], wsum[NBINS_MAX / 64];
  uint32_t* stage_val = reinterpret_cast<uint32_t*>(h2_msm_smem);             // P1_TS * W payloads
  uint16_t* stage_key = reinterpret_cast<uint16_t*>(stage_val + P1_TS * W);   // P1_TS * W bucket ids
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
  uint32_t gb = 0;
  if (tid < NBINS_MAX) {
    cnt[tid] = 0;
    gb = tid < nbins ? base[(size_t)tid * ntiles + tile] : 0;  // this tile's first slot in bin `tid`
  }
  __syncthreads();
  size_t i = (size_t)tile * P1_TS + tid;
  fe s;
  const bool live = i < n + (shift ? 1 : 0);
  // compile-time windows: the rank each entry drew from its bin counter is kept in registers, so the digits
  // are walked once; run-time c walks them twice (count, then place) to stay out of scratch arrays
  constexpr uint32_t WK = CT ? (255 + CT - 1) / CT : 1;
  uint32_t ent[WK], rk[WK];
#pragma unroll
  for (uint32_t w = 0; w < WK; w++) ent[w] = 0xFFFFFFFFu;
  bool live_nz = false;
  if (live) {
    s = msm_scalar(scalars, i, n, shift);
    live_nz = !fe_is_zero(s);
  }
  if (live_nz) {
    for_each_digit<CT>(s, c, W, [&](uint32_t w, uint32_t bucket, uint32_t neg) {
      uint32_t r = atomicAdd(&cnt[bin_of<WIDE>(bucket, lb)], 1u);
      if (CT) {
        ent[w] = bucket | (neg << 31);
        rk[w] = r;
      }
    });
  }
  __syncthreads();
  // exclusive scan of the bin counts: lstart[b] = first staging slot of bin b, cnt[b] = running cursor
  uint32_t v = 0, inc = 0;
  if (tid < NBINS_MAX) {
    v = cnt[tid];
    inc = wave_incl_scan(v);
    if (lane == 63) wsum[wave] = inc;
  }
  __syncthreads();
  if (tid < NBINS_MAX) {
    uint32_t add = 0;
    for (uint32_t j = 0; j < wave; j++) add += wsum[j];
    lstart[tid] = add + inc - v;
    cnt[tid] = add + inc - v;
    if (tid == NBINS_MAX - 1) lstart[NBINS_MAX] = add + inc;
  }
  __syncthreads();
  const uint32_t mask = (1u << lb) - 1;
  if (CT) {
#pragma unroll
    for (uint32_t w = 0; w < WK; w++)
      if (ent[w] != 0xFFFFFFFFu) {
        const uint32_t bucket = ent[w] & 0x7FFFFFFFu;
        const uint32_t pos = lstart[bin_of<WIDE>(bucket, lb)] + rk[w];
        stage_val[pos] = (ent[w] & 0x80000000u) | (uint32_t)((size_t)w * n_reg + i);
        stage_key[pos] = (uint16_t)(WIDE ? key_of<true>(bucket, mask) : bucket);
      }
  } else if (live_nz) {
    for_each_digit<CT>(s, c, W, [&](uint32_t w, uint32_t bucket, uint32_t neg) {
      uint32_t pos = atomicAdd(&cnt[bucket >> lb], 1u);
      stage_val[pos] = (neg << 31) | (uint32_t)((size_t)w * n_reg + i);
      stage_key[pos] = (uint16_t)bucket;
    });
  }
  __syncthreads();
  // each wave writes whole bins: one contiguous run per (bin, tile)
  // the staged pairs are grouped by bin: slot p of bin b goes to base[b][tile] + (p - lstart[b]); consecutive
  // slots are consecutive in HBM inside a (bin, tile) run, so a wavefront's stores coalesce
  if (tid < NBINS_MAX) cnt[tid] = gb - lstart[tid];
  __syncthreads();
  const uint32_t total = lstart[NBINS_MAX];
  if (WIDE) {  // the staged key no longer names the bin: it is the one whose run [lstart[b], lstart[b+1]) holds slot p
    uint16_t* keys_out = reinterpret_cast<uint16_t*>(keys_out_);
    for (uint32_t p = tid; p < total; p += P1_TS) {
      uint32_t lo = 0, hi = NBINS_MAX;  // invariant: lstart[lo] <= p < lstart[hi]
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (lstart[mid] <= p) lo = mid;
        else hi = mid;
      }
      const uint32_t dst = p + cnt[lo];
      vals_out[dst] = stage_val[p];
      keys_out[dst] = stage_key[p];
    }
  } else {
    uint8_t* keys_out = reinterpret_cast<uint8_t*>(keys_out_);
    for (uint32_t p = tid; p < total; p += P1_TS) {
      const uint32_t key = stage_key[p];
      const uint32_t dst = p + cnt[key >> lb];
      vals_out[dst] = stage_val[p];
      keys_out[dst] = (uint8_t)(key & mask);
    }
  }
}
template <uint32_t CT>
__global__ void __launch_bounds__(P1_TS) k_msm_bin_scatter(const fe* scalars, size_t n, const fe* shift, size_t n_reg, uint32_t c, uint32_t W,
                                                           uint32_t lb, uint32_t nbins, uint32_t ntiles, const uint32_t* base, uint32_t* vals_out,
                                                           void* keys_out_, const uint32_t* tile_live) {
  msm_bin_scatter_body<CT>(scalars, n, shift, n_reg, c, W, lb, nbins, ntiles, base, vals_out, keys_out_, tile_live);
}

constexpr uint32_t NQ_MAX = 128;  // buckets per bin (c = 17: 2^16 buckets in 512 bins)
__device__ __forceinline__ void msm_bin_sort_body(const uint8_t* keys_in, const uint32_t* vals_in, const uint32_t* base, uint32_t ntiles, uint32_t nbins,
                                                  uint32_t lb, uint32_t s0_fixed, uint32_t nb, uint32_t* vals_out, uint32_t* off, uint32_t* np0,
                                                  uint32_t* np1, uint32_t* s0_out) {
  H2_AB_PRIO();
  __shared__ uint32_t wh[P2_THREADS / 64][NQ_MAX];
  __shared__ uint32_t run[NQ_MAX], ccnt[NQ_MAX], cstart[NQ_MAX], carry64;
  __shared__ uint32_t stage[P2_CH];
  __shared__ uint8_t stage_q[P2_CH];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, bin = blockIdx.x;
  const uint32_t start = base[(size_t)bin * ntiles], end = base[(size_t)(bin + 1) * ntiles];
  const uint32_t nq = 1u << lb;
  const uint32_t entries = base[(size_t)nbins * ntiles];
  const uint32_t s0 = s0_fixed ? s0_fixed : accum_chunk_len(entries);  // every workgroup derives the same value
  for (uint32_t j = tid; j < (P2_THREADS / 64) * NQ_MAX; j += P2_THREADS) (&wh[0][0])[j] = 0;
  __syncthreads();
  // 16 keys per load; the buffer is padded so that the aligned window may overhang [start, end)
  for (uint32_t j = (start & ~15u) + tid * 16; j < end; j += P2_THREADS * 16) {
    const uint4 kk = *reinterpret_cast<const uint4*>(keys_in + j);
    const uint32_t w4[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) {
      const uint32_t idx = j + t;
      if (idx >= start && idx < end) atomicAdd(&wh[wave][(w4[t >> 2] >> (8 * (t & 3))) & (NQ_MAX - 1)], 1u);
    }
  }
  __syncthreads();
  // exclusive scan over the bin's buckets (two wavefronts of 64)
  uint32_t tot = 0, inc = 0;
  if (tid < NQ_MAX) {
#pragma unroll
    for (uint32_t w = 0; w < P2_THREADS / 64; w++) tot += wh[w][tid];
    inc = wave_incl_scan(tot);
    if (tid == 63) carry64 = inc;
  }
  __syncthreads();
  if (tid < NQ_MAX) {
    const uint32_t ex = inc - tot + (tid >= 64 ? carry64 : 0);
    run[tid] = start + ex;
    ccnt[tid] = 0;
    if (tid < nq) {
      const uint32_t b = (bin << lb) + tid;
      const uint32_t o = start + ex;
      off[b] = o;
      // partial sums the accumulation leaves for this bucket: one, plus one per chunk start inside its run
      uint32_t f0 = tot ? 1u + (o + tot - 1) / s0 - o / s0 : 0u;
      np0[b] = f0;
      np1[b] = (f0 + S1 - 1) / S1;
    }
    if (bin == 0 && tid == 0) {  // entry nb: end of the sorted array; the task scans leave their totals there
      off[nb] = entries;
      np0[nb] = 0;
      np1[nb] = 0;
      *s0_out = s0;
    }
  }
  __syncthreads();
  for (uint32_t cs = start; cs < end; cs += P2_CH) {
    const uint32_t m = min(P2_CH, end - cs);
    uint32_t q[P2_PER], v[P2_PER], r[P2_PER];
#pragma unroll
    for (uint32_t k = 0; k < P2_PER; k++) {
      const uint32_t p = k * P2_THREADS + tid;
      if (p < m) {
        q[k] = keys_in[cs + p] & (NQ_MAX - 1);
        v[k] = vals_in[cs + p];
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < P2_PER; k++)
      if (k * P2_THREADS + tid < m) r[k] = atomicAdd(&ccnt[q[k]], 1u);
    __syncthreads();
    uint32_t cn = 0, cinc = 0;
    if (tid < NQ_MAX) {
      cn = ccnt[tid];
      cinc = wave_incl_scan(cn);
      if (tid == 63) carry64 = cinc;
    }
    __syncthreads();
    if (tid < NQ_MAX) cstart[tid] = cinc - cn + (tid >= 64 ? carry64 : 0);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < P2_PER; k++)
      if (k * P2_THREADS + tid < m) {
        const uint32_t pos = cstart[q[k]] + r[k];
        stage[pos] = v[k];
        stage_q[pos] = (uint8_t)q[k];
      }
    __syncthreads();
    for (uint32_t p = tid; p < m; p += P2_THREADS) {
      const uint32_t qq = stage_q[p];
      vals_out[run[qq] + (p - cstart[qq])] = stage[p];
    }
    __syncthreads();
    if (tid < NQ_MAX) {
      run[tid] += ccnt[tid];
      ccnt[tid] = 0;
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(P2_THREADS) k_msm_bin_sort(const uint8_t* keys_in, const uint32_t* vals_in, const uint32_t* base, uint32_t ntiles,
                                                             uint32_t nbins, uint32_t lb, uint32_t s0_fixed, uint32_t nb, uint32_t* vals_out,
                                                             uint32_t* off, uint32_t* np0, uint32_t* np1, uint32_t* s0_out) {
  msm_bin_sort_body(keys_in, vals_in, base, ntiles, nbins, lb, s0_fixed, nb, vals_out, off, np0, np1, s0_out);
}

// k_msm_bin_sort for wide windows: up to NQW = 1024 buckets per bin, 16-bit keys, one shared histogram (1024 counters
// see little contention), two counters per thread in the scans, chunks of 6144 payloads staged in LDS (52 KB in all).
constexpr uint32_t NQW = 1024, P2W_PER = 12, P2W_CH = P2_THREADS * P2W_PER;
__device__ __forceinline__ void block_excl_scan_1024(const uint32_t* in, uint32_t* out, uint32_t* wtot /* P2_THREADS / 64 */) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t a = in[2 * tid], b = in[2 * tid + 1];
  const uint32_t inc = wave_incl_scan(a + b);
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  uint32_t add = 0;
  for (uint32_t w = 0; w < wave; w++) add += wtot[w];
  const uint32_t ex = add + inc - (a + b);
  __syncthreads();  // wtot is reused by the next scan
  out[2 * tid] = ex;
  out[2 * tid + 1] = ex + a;
}
__global__ void __launch_bounds__(P2_THREADS) k_msm_bin_sort_wide(const uint16_t* keys_in, const uint32_t* vals_in, const uint32_t* base,
                                                                  uint32_t ntiles, uint32_t nbins, uint32_t lb, uint32_t s0_fixed, uint32_t nb,
                                                                  uint32_t* vals_out, uint32_t* off, uint32_t* np0, uint32_t* np1, uint32_t* s0_out) {
  __shared__ uint32_t hist[NQW], run[NQW], ccnt[NQW], cstart[NQW], wtot[P2_THREADS / 64];
  __shared__ uint32_t stage[P2W_CH];
  __shared__ uint16_t stage_q[P2W_CH];
  const uint32_t tid = threadIdx.x, bin = blockIdx.x;
  const uint32_t start = base[(size_t)bin * ntiles], end = base[(size_t)(bin + 1) * ntiles];
  const uint32_t nq = 1u << lb;
  const uint32_t entries = base[(size_t)nbins * ntiles];
  const uint32_t s0 = s0_fixed ? s0_fixed : accum_chunk_len(entries);
  for (uint32_t j = tid; j < NQW; j += P2_THREADS) {
    hist[j] = 0;
    ccnt[j] = 0;
  }
  __syncthreads();
  // 8 keys per load; the buffer is padded so that the aligned window may overhang [start, end)
  for (uint32_t j = (start & ~7u) + tid * 8; j < end; j += P2_THREADS * 8) {
    const uint4 kk = *reinterpret_cast<const uint4*>(keys_in + j);
    const uint32_t w4[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) {
      const uint32_t idx = j + t;
      if (idx >= start && idx < end) atomicAdd(&hist[(w4[t >> 1] >> (16 * (t & 1))) & (NQW - 1)], 1u);
    }
  }
  __syncthreads();
  block_excl_scan_1024(hist, cstart, wtot);
  __syncthreads();
  for (uint32_t q = tid; q < NQW; q += P2_THREADS) {
    const uint32_t ex = cstart[q], tot = hist[q];
    run[q] = start + ex;
    if (q < nq) {
      const uint32_t b = (bin << lb) + q;
      const uint32_t o = start + ex;
      off[b] = o;
      const uint32_t f0 = tot ? 1u + (o + tot - 1) / s0 - o / s0 : 0u;
      np0[b] = f0;
      np1[b] = (f0 + S1 - 1) / S1;
    }
  }
  if (bin == 0 && tid == 0) {
    off[nb] = entries;
    np0[nb] = 0;
    np1[nb] = 0;
    *s0_out = s0;
  }
  __syncthreads();
  for (uint32_t cs = start; cs < end; cs += P2W_CH) {
    const uint32_t m = min(P2W_CH, end - cs);
    uint32_t q[P2W_PER], v[P2W_PER], r[P2W_PER];
#pragma unroll
    for (uint32_t k = 0; k < P2W_PER; k++) {
      const uint32_t p = k * P2_THREADS + tid;
      if (p < m) {
        q[k] = keys_in[cs + p] & (NQW - 1);
        v[k] = vals_in[cs + p];
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < P2W_PER; k++)
      if (k * P2_THREADS + tid < m) r[k] = atomicAdd(&ccnt[q[k]], 1u);
    __syncthreads();
    block_excl_scan_1024(ccnt, cstart, wtot);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < P2W_PER; k++)
      if (k * P2_THREADS + tid < m) {
        const uint32_t pos = cstart[q[k]] + r[k];
        stage[pos] = v[k];
        stage_q[pos] = (uint16_t)q[k];
      }
    __syncthreads();
    for (uint32_t p = tid; p < m; p += P2_THREADS) {
      const uint32_t qq = stage_q[p];
      vals_out[run[qq] + (p - cstart[qq])] = stage[p];
    }
    __syncthreads();
    for (uint32_t qi = tid; qi < NQW; qi += P2_THREADS) {
      run[qi] += ccnt[qi];
      ccnt[qi] = 0;
    }
    __syncthreads();
  }
}

// Partial bucket sums travel between the MSM kernels as raw xyzz29 values (4 x 9 normalized limbs =
// 144 B, Montgomery-2^261, loosely reduced): no conversion until the final result.
constexpr uint32_t PART_BYTES = 144;
__device__ __forceinline__ xyzz29 part_load(const uint8_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint4 v = q[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  xyzz29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) { r.x.v[i] = w[i]; r.y.v[i] = w[9 + i]; r.zz.v[i] = w[18 + i]; r.zzz.v[i] = w[27 + i]; }
  return r;
}
__device__ __forceinline__ void part_store(uint8_t* p, const xyzz29& a) {
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) { w[i] = a.x.v[i]; w[9 + i] = a.y.v[i]; w[18 + i] = a.zz.v[i]; w[27 + i] = a.zzz.v[i]; }
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// largest b with toff[b] <= t (toff has nb+1 monotone entries, toff[nb] = total > t)
__device__ __forceinline__ uint32_t find_bucket(const uint32_t* toff, uint32_t nb, uint32_t t) {
  uint32_t lo = 0, hi = nb;  // invariant: toff[lo] <= t < toff[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (toff[mid] <= t) lo = mid;
    else hi = mid;
  }
  return lo;
}

// what the accumulation gathers per entry: the 64-byte table entry (two canonical 8-word values, unpacked to 9 limbs each after
// the load).  A 72-byte pre-unpacked entry (~36 fewer instructions per addition of ~2320, 12 % more gathered bytes) was built and
// measured in round 4 (commit 51bbdca, profiles/r04_accum_ab.txt): replay step +1 .. 2 %, not kept.
typedef affine tab_entry;
__device__ __forceinline__ tab_entry tab_load(const uint8_t* table, uint32_t i) { return affine_load(table + (size_t)i * 64); }
__device__ __forceinline__ bool tab_is_identity(const tab_entry& p) { return affine_is_identity(p); }
__device__ __forceinline__ f29 tab_x(const tab_entry& p) { return f29_unpack(p.x.v); }
__device__ __forceinline__ f29 tab_y(const tab_entry& p) { return f29_unpack(p.y.v); }

// level 0: the bucket-ordered entry array is cut into chunks of exactly s0 entries (any value), one thread per
// chunk, regardless of bucket boundaries: every lane of a wavefront performs the same number of additions
// (cutting each bucket into its own tasks left a short remainder task per bucket: ~7 % idle lanes at 512
// entries per bucket).  A chunk that crosses a bucket boundary closes one partial sum and opens the next;
// partial sums are numbered in array order, so bucket b owns np0[b] = 1 + (#chunk starts strictly inside
// its run) consecutive partials starting at toff[b] (the exclusive scan of np0) — the layout the fold
// expects.  Gathers table points (64 B), mixed additions in the lazy 29-bit-limb representation.
__device__ __forceinline__ void msm_accum_body(const uint32_t* entries, const uint32_t* off, const uint32_t* toff, uint32_t nb, const uint32_t* s0_dev,
                                               const uint8_t* table, uint8_t* part) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = off[nb];
  const uint32_t s0 = *s0_dev;
  if ((uint64_t)t * s0 >= total) return;
  const uint32_t start = t * s0;
  const uint32_t end = min(start + s0, total);
  uint32_t b = find_bucket(off, nb, start);  // the (non-empty) bucket that holds entry `start`
  uint32_t pidx = toff[b] + (t - off[b] / s0);
  uint32_t bend = off[b + 1];
  xyzz29 acc = xyzz29_identity();
  uint32_t e = entries[start];
  tab_entry nxt = tab_load(table, e & 0x7fffffffu);
  uint32_t nneg = e >> 31;
  for (uint32_t k = start; k < end; k++) {
    if (k == bend) {  // bucket boundary inside the chunk: close this partial, move to the next non-empty bucket
      part_store(part + (size_t)pidx * PART_BYTES, acc);
      pidx++;
      acc = xyzz29_identity();
      do {
        b++;
        bend = off[b + 1];
      } while (bend == k);
    }
    tab_entry p = nxt;
    uint32_t neg = nneg;
    if (k + 1 < end) {
      e = entries[k + 1];
      nxt = tab_load(table, e & 0x7fffffffu);
      nneg = e >> 31;
    }
    if (tab_is_identity(p)) continue;
    f29 x2 = tab_x(p);
    f29 y2 = tab_y(p);
    if (neg) y2 = f29_sub(f29_zero(), y2, Fq29::K2);  // 2p - y (lazy)
    xyzz29_madd(acc, x2, y2);
  }
  part_store(part + (size_t)pidx * PART_BYTES, acc);
}
__global__ void __launch_bounds__(256) k_msm_accum(const uint32_t* entries, const uint32_t* off, const uint32_t* toff, uint32_t nb,
                                                    const uint32_t* s0_dev, const uint8_t* table, uint8_t* part) {
  msm_accum_body(entries, off, toff, nb, s0_dev, table, part);
}

// ---- pair-affine accumulation (round 5 experiment; selected with H2MI_MSM_PA in the -DH2MI_AB library only) ------------------------
// The accumulation spends 1467 multiply-adds per entry on a mixed XYZZ addition.  Two table points of the same bucket can instead be
// added in AFFINE coordinates first — 5 multiplications and a squaring (936 multiply-adds) once the inverse of x2 - x1 is known — and
// only their sum enters the accumulator: 2403 instead of 2934 multiply-adds per pair (-18 %).  The inverses are shared by EVERY pair of
// the launch (Montgomery's trick in two levels), which takes three kernels:
//   k_msm_pa_forward   per chunk (the accumulation's chunks): running product of the pairs' x2 - x1, the product BEFORE each pair
//                      spilled (36 B per pair, lane-contiguous), the chunk's total written out.  Gathers every table point's x.
//   k_msm_pa_invert    the inverses of the chunk totals: per workgroup 1024 totals, prefix / suffix products, ONE division-step inversion
//   k_msm_pa_backward  the accumulation itself, walking its chunk BACKWARDS (the order in which Montgomery's trick releases the
//                      inverses): pair sum in affine coordinates, then one mixed addition; same partial sums, same layout as
//                      k_msm_accum (partials are numbered in array order, so walking down decrements the index)
// Pairs are (start + 2j, start + 2j + 1) of a chunk; a pair that straddles a bucket boundary, holds an identity or two points with
// the same x (P + P, P - P) is not combined: both kernels decide that from the same data and the backward pass adds such entries one
// by one, exactly as k_msm_accum does.
// Price: every table point is gathered twice and 72 B per pair are spilled and read back — 2.6 GB instead of 0.9 GB per 2^20 MSM.
// MEASURED (profiles/r05_pair_affine_ab.txt, tools/pa_ab.py): a loss everywhere.  2^20: forward 0.46 ms + inversions 0.09 ms, and the
// backward pass itself takes 1.31 ms against k_msm_accum's 1.11 ms although it issues 18 % fewer multiply-adds — two gathers and a
// spill read per pair at 256 VGPRs leave the latency uncovered; back to back 1.76 against 1.27 ms per MSM, alone 2.34 against 1.62.
// Kept in the A/B library (make ab) as the measured form of the estimate in HISTORY.md; the product does not contain it.
#ifdef H2MI_AB
constexpr uint32_t PA_MAX_PAIRS = S0_MAX / 2;
__device__ __forceinline__ bool words_equal8(const uint32_t* a, const uint32_t* b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d |= a[i] ^ b[i];
  return d == 0;
}
__device__ __forceinline__ bool words_zero8(const uint32_t* a) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d |= a[i];
  return d == 0;
}
// can the two table points be added by the affine formula?  (decided identically by the forward and the backward pass)
__device__ __forceinline__ bool pa_pair_ok(const tab_entry& p0, const tab_entry& p1) {
  return !words_zero8(p0.x.v) && !words_zero8(p1.x.v) && !words_equal8(p0.x.v, p1.x.v);
}
__device__ __forceinline__ void pa_store(uint32_t* base, uint32_t T, uint32_t t, const f29& a) {
#pragma unroll
  for (int i = 0; i < 9; i++) base[(size_t)i * T + t] = a.v[i];
}
__device__ __forceinline__ f29 pa_load(const uint32_t* base, uint32_t T, uint32_t t) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = base[(size_t)i * T + t];
  return r;
}
__global__ void __launch_bounds__(256) k_msm_pa_forward(const uint32_t* entries, const uint32_t* off, uint32_t nb, const uint32_t* s0_dev,
                                                         const uint8_t* table, uint32_t T, uint32_t* spill, uint32_t* tot) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const uint32_t total = off[nb];
  const uint32_t s0 = *s0_dev;
  f29 prod = f29_const<Fq29>(Fq29::ONE);
  if ((uint64_t)t * s0 < total && s0 <= 2 * PA_MAX_PAIRS) {
    const uint32_t start = t * s0;
    const uint32_t end = min(start + s0, total);
    uint32_t b = find_bucket(off, nb, start);
    uint32_t bend = off[b + 1];
    tab_entry n0, n1;
    if (start + 1 < end) {
      n0 = tab_load(table, entries[start] & 0x7fffffffu);
      n1 = tab_load(table, entries[start + 1] & 0x7fffffffu);
    }
    for (uint32_t k = start, j = 0; k + 1 < end; k += 2, j++) {
      const tab_entry p0 = n0, p1 = n1;
      if (k + 3 < end) {  // the next pair's points travel while this pair's product is formed
        n0 = tab_load(table, entries[k + 2] & 0x7fffffffu);
        n1 = tab_load(table, entries[k + 3] & 0x7fffffffu);
      }
      while (k >= bend) {
        b++;
        bend = off[b + 1];
      }
      if (k + 1 >= bend || !pa_pair_ok(p0, p1)) continue;
      pa_store(spill + (size_t)j * 9 * T, T, t, prod);
      prod = f29_mul<Fq29>(prod, affine29_pair_diff(tab_x(p0), tab_x(p1)));
    }
  }
  pa_store(tot, T, t, prod);
}
// tot[t] <- 1 / tot[t] for t < T: 1024 values per workgroup of 256 threads
__global__ void __launch_bounds__(256) k_msm_pa_invert(uint32_t* tot, uint32_t T) {
  __shared__ f29 pre[256], suf[256];
  __shared__ f29 winv;
  const uint32_t tid = threadIdx.x, base = blockIdx.x * 1024;
  f29 v[4], q[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t idx = base + i * 256 + tid;
    v[i] = idx < T ? pa_load(tot, T, idx) : f29_const<Fq29>(Fq29::ONE);
    q[i] = i ? f29_mul<Fq29>(q[i - 1], v[i]) : v[i];
  }
  pre[tid] = q[3];
  suf[tid] = q[3];
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {  // inclusive prefix products in pre, inclusive suffix products in suf
    f29 a, c;
    if (tid >= d) a = pre[tid - d];
    if (tid + d < 256) c = suf[tid + d];
    __syncthreads();
    if (tid >= d) pre[tid] = f29_mul<Fq29>(pre[tid], a);
    if (tid + d < 256) suf[tid] = f29_mul<Fq29>(suf[tid], c);
    __syncthreads();
  }
  if (tid < 64) {  // every lane of the first wavefront on the same value: uniform branches.  (W 2^261) read as Montgomery-2^256 is
                   // (32 W) 2^256; its inverse times 2^10 is W^-1 2^261
    fe w;
    f29_pack(f29_reduce_canonical<Fq29>(pre[255]), w.v);
    fe inv = fe_inv_ds<Fq>(w);
    for (int i = 0; i < 10; i++) inv = fe_dbl<Fq>(inv);
    if (tid == 0) winv = f29_unpack(inv.v);
  }
  __syncthreads();
  f29 r = winv;  // -> 1 / q[3] of this thread: the workgroup's inverse times everyone else's totals
  if (tid) r = f29_mul<Fq29>(r, pre[tid - 1]);
  if (tid < 255) r = f29_mul<Fq29>(r, suf[tid + 1]);
#pragma unroll
  for (int i = 3; i >= 0; i--) {
    const uint32_t idx = base + i * 256 + tid;
    const f29 inv_i = i ? f29_mul<Fq29>(r, q[i - 1]) : r;
    if (i) r = f29_mul<Fq29>(r, v[i]);
    if (idx < T) pa_store(tot, T, idx, inv_i);
  }
}
__global__ void __launch_bounds__(256) k_msm_pa_backward(const uint32_t* entries, const uint32_t* off, const uint32_t* toff, uint32_t nb,
                                                          const uint32_t* s0_dev, const uint8_t* table, uint8_t* part, uint32_t T,
                                                          const uint32_t* spill, const uint32_t* totinv) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = off[nb];
  const uint32_t s0 = *s0_dev;
  if (t >= T || (uint64_t)t * s0 >= total) return;
  const uint32_t start = t * s0;
  const uint32_t end = min(start + s0, total);
  const bool pairs_on = s0 <= 2 * PA_MAX_PAIRS;  // longer chunks (a chunk override): the forward pass combined nothing
  uint32_t b = find_bucket(off, nb, end - 1);  // the bucket that holds the chunk's LAST entry
  uint32_t bstart = off[b];
  uint32_t pidx = toff[b] + (t - off[b] / s0);
  xyzz29 acc = xyzz29_identity();
  f29 run = pa_load(totinv, T, t);
  // make `idx` (below every entry handled so far) an entry of the current bucket: crossing a boundary closes a partial sum
  auto cross = [&](uint32_t idx) {
    if (idx >= bstart) return;
    part_store(part + (size_t)pidx * PART_BYTES, acc);
    pidx--;
    acc = xyzz29_identity();
    do {
      b--;
    } while (off[b] > idx);
    bstart = off[b];
  };
  auto single = [&](const tab_entry& p, uint32_t neg) {
    if (tab_is_identity(p)) return;
    f29 y = tab_y(p);
    if (neg) y = f29_sub(f29_zero(), y, Fq29::K2);
    xyzz29_madd(acc, tab_x(p), y);
  };
  uint32_t hi = end;  // entries [hi, end) are done
  if ((end - start) & 1u) {
    const uint32_t e = entries[end - 1];
    single(tab_load(table, e & 0x7fffffffu), e >> 31);
    hi = end - 1;
  }
  uint32_t e0 = 0, e1 = 0;
  tab_entry n0, n1;
  if (hi > start) {
    e0 = entries[hi - 2];
    e1 = entries[hi - 1];
    n0 = tab_load(table, e0 & 0x7fffffffu);
    n1 = tab_load(table, e1 & 0x7fffffffu);
  }
  for (uint32_t k1 = hi; k1 > start; k1 -= 2) {  // the pair (k1 - 2, k1 - 1), j = (k1 - 2 - start) / 2
    const uint32_t k0 = k1 - 2, j = (k0 - start) >> 1;
    const tab_entry p0 = n0, p1 = n1;
    const uint32_t neg0 = e0 >> 31, neg1 = e1 >> 31;
    if (k0 > start) {
      e0 = entries[k0 - 2];
      e1 = entries[k0 - 1];
      n0 = tab_load(table, e0 & 0x7fffffffu);
      n1 = tab_load(table, e1 & 0x7fffffffu);
    }
    cross(k0 + 1);
    if (pairs_on && k0 >= bstart && pa_pair_ok(p0, p1)) {
      const f29 before = pa_load(spill + (size_t)j * 9 * T, T, t);
      const f29 x0 = tab_x(p0), x1 = tab_x(p1);
      const f29 dinv = f29_mul<Fq29>(run, before);
      run = f29_mul<Fq29>(run, affine29_pair_diff(x0, x1));
      f29 x3, y3;
      affine29_pair_add(x0, tab_y(p0), neg0 != 0, x1, tab_y(p1), neg1 != 0, dinv, x3, y3);
      xyzz29_madd(acc, x3, y3);
    } else {
      single(p1, neg1);
      cross(k0);
      single(p0, neg0);
    }
  }
  part_store(part + (size_t)pidx * PART_BYTES, acc);
}
#endif  // H2MI_AB

// ---- batched head: the partition and the accumulation of up to HEAD_BATCH MSMs of one length over ONE base set as ONE set of launches
// (blockIdx.y = MSM; round 4).  At 2^16 rows and below a prover phase's three or four commitments were issued at the HOST's pace: eight
// launches and six event operations per MSM, ~77 us each on the compiled host, while the kernels themselves run 4 - 13 us (proof
// timeline at 2^16: the three z commitments' partitions occupy 270 us of the phase one after the other).  The kernels are the
// single-MSM bodies behind a descriptor; h2mi_msm_bn254_g1_phase_dev chooses this form up to HEAD_BATCH_MAX_N points — beyond, the
// stage-by-stage pipelining of consecutive MSMs (partition of one beside the accumulation of another) is worth more than the launches.
constexpr uint32_t HEAD_BATCH = 4;
struct HeadDesc {
  const fe* scalars;
  fe* shift;  // nullptr: no dominant-value shift
  uint32_t *bincnt, *tile_live, *binseg, *binbase, *vals0, *vals1, *off, *np0, *np1, *toff0, *toff1, *s0_dev;
  void* bkeys;
  uint8_t* part0;
};
struct HeadBatch {
  HeadDesc d[HEAD_BATCH];
};
__global__ void __launch_bounds__(64) k_msm_pick_shift_b(const HeadBatch hb, size_t n) {
  const HeadDesc& d = hb.d[blockIdx.y];
  msm_pick_shift_body(d.scalars, n, d.shift);
}
template <uint32_t CT>
__global__ void __launch_bounds__(P1_TS) k_msm_bin_count_b(const HeadBatch hb, size_t n, uint32_t c, uint32_t W, uint32_t lb, uint32_t nbins,
                                                           uint32_t ntiles) {
  const HeadDesc& d = hb.d[blockIdx.y];
  msm_bin_count_body<CT>(d.scalars, n, d.shift, c, W, lb, nbins, ntiles, d.bincnt, d.tile_live);
}
template <uint32_t CT>
__global__ void __launch_bounds__(P1_TS) k_msm_bin_scatter_b(const HeadBatch hb, size_t n, size_t n_reg, uint32_t c, uint32_t W, uint32_t lb,
                                                             uint32_t nbins, uint32_t ntiles) {
  const HeadDesc& d = hb.d[blockIdx.y];
  msm_bin_scatter_body<CT>(d.scalars, n, d.shift, n_reg, c, W, lb, nbins, ntiles, d.binbase, d.vals0, d.bkeys, d.tile_live);
}
__global__ void __launch_bounds__(P2_THREADS) k_msm_bin_sort_b(const HeadBatch hb, uint32_t ntiles, uint32_t nbins, uint32_t lb, uint32_t s0_fixed,
                                                               uint32_t nb) {
  const HeadDesc& d = hb.d[blockIdx.y];
  msm_bin_sort_body((const uint8_t*)d.bkeys, d.vals0, d.binbase, ntiles, nbins, lb, s0_fixed, nb, d.vals1, d.off, d.np0, d.np1, d.s0_dev);
}
__global__ void __launch_bounds__(256) k_msm_accum_b(const HeadBatch hb, uint32_t nb, const uint8_t* table) {
  const HeadDesc& d = hb.d[blockIdx.y];
  msm_accum_body(d.vals1, d.off, d.toff0, nb, d.s0_dev, table, d.part0);
}
// the scans of scan.cuh over the batch's arrays: blockIdx.y = MSM (bins), blockIdx.y = array and blockIdx.z = MSM (task counts)
__global__ void __launch_bounds__(1024) k_scan_segsum_bins_b(const HeadBatch hb, uint32_t m) {
  const HeadDesc& d = hb.d[blockIdx.y];
  scan_segsum_body<SCAN_SEG_BINS>(d.bincnt, m, d.binseg);
}
__global__ void __launch_bounds__(1024) k_scan_seg_bins_b(const HeadBatch hb, uint32_t m) {
  const HeadDesc& d = hb.d[blockIdx.y];
  scan_seg_body<SCAN_SEG_BINS>(d.bincnt, d.binbase, m, d.binseg);
}
__global__ void __launch_bounds__(1024) k_scan_seg_tasks_b(const HeadBatch hb, uint32_t m) {
  const HeadDesc& d = hb.d[blockIdx.z];
  scan_seg_body<SCAN_SEG_TASKS>(blockIdx.y ? d.np1 : d.np0, blockIdx.y ? d.toff1 : d.toff0, m, nullptr);
}

// ---- bucket reduction ("tail"), batched --------------------------------------------------------------
// fold -> finish -> rowcol -> weighted -> final is a chain of ~40 dependent point operations (5-9 us each
// for a lone wavefront) over little data, so its cost is latency, not throughput.  MSMs queued on the
// library stream therefore only leave their accumulated partials behind; the tails of all MSMs since the
// last join run as ONE batch of these kernels (blockIdx.y = MSM), paying the latency once per batch.
constexpr uint32_t TAIL_BATCH = 8;
struct TailDesc {
  const uint8_t* part0;   // accumulation output
  const uint32_t *toff0, *np0, *toff1, *np1, *off;
  uint8_t *part1, *dense, *rc, *g, *out;
  uint64_t* stats;
  uint32_t nb, logNh, logNl, canonical;
  // wide windows: dense holds nb = 2^(MAT_LOG + seg_log) buckets; k_msm_seg leaves the segment sums in mat (what rowcol
  // reads; = dense when seg_log = 0) and the segments' local weighted sums in vsum
  const uint8_t* mat;
  uint8_t *dense2, *vsum;
  uint32_t seg_log;
  uint32_t hot_min;  // a bucket with more partials than this is finished by a whole workgroup (k_msm_finish_hot)
  uint32_t no_fold;  // the finish kernels read the accumulation's partials directly (part1 / toff1 / np1 alias part0 / toff0 / np0)
};
struct TailBatch {
  TailDesc d[TAIL_BATCH];
};

// Every point operation below is lane-cooperative (g1_29_quad.cuh): the four lanes of a quad hold the same
// operands and compute one addition / doubling together in 4 / 3 multiplication rounds instead of 14 / 10
// dependent multiplications, which is what shortens the chain.  "quad q" = lanes 4q .. 4q+3.
constexpr uint32_t TAIL_THREADS = 512;  // rowcol / weighted: 128 quads reduce up to 256 values

// fold level: one worker per task of <= S1 partials of one bucket.  A worker is a quad (QUAD: small grids,
// where the chain's latency is what counts) or a single lane (large grids, where the fold is bound by
// throughput and the cooperative form's ~2x instruction count would cost more than its shorter chain saves).
template <bool QUAD>
__global__ void __launch_bounds__(256) k_msm_fold(const TailBatch tb) {
  const TailDesc& d = tb.d[blockIdx.y];
  if (d.no_fold) return;  // wide windows and mid-size base sets skip the fold level (tail_desc)
  const uint32_t nb = d.nb;
  const uint32_t *toff_out = d.toff1, *toff_in = d.toff0, *np_in = d.np0;
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> (QUAD ? 2 : 0);
  if (t >= toff_out[nb]) return;
  uint32_t b = find_bucket(toff_out, nb, t);
  uint32_t j = t - toff_out[b];
  const uint32_t m = np_in[b], np = toff_out[b + 1] - toff_out[b];  // balanced split of the bucket's partials
  const uint32_t q = m / np, r = m - q * np;
  uint32_t start = toff_in[b] + j * q + min(j, r);
  uint32_t len = q + (j < r ? 1u : 0u);
  xyzz29 acc = part_load(d.part0 + (size_t)start * PART_BYTES);
  for (uint32_t k = 1; k < len; k++) {
    xyzz29 p = part_load(d.part0 + (size_t)(start + k) * PART_BYTES);
    if (QUAD) acc = xyzz29_add_quad(acc, p);
    else xyzz29_add(acc, p);
  }
  if (!QUAD || (threadIdx.x & 3u) == 0) part_store(d.part1 + (size_t)t * PART_BYTES, acc);
}

// bucket finish: FG workers (quads or lanes, as above) cooperate on one bucket: worker l sums partials l,
// l+FG, ... (one partial each in the common case), then a log2(FG)-level shuffle tree; writes one dense XYZZ
// value per bucket.
__device__ __forceinline__ xyzz29 shfl_down_xyzz(const xyzz29& a, uint32_t delta) {
  xyzz29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.x.v[i] = __shfl_down(a.x.v[i], delta);
    r.y.v[i] = __shfl_down(a.y.v[i], delta);
    r.zz.v[i] = __shfl_down(a.zz.v[i], delta);
    r.zzz.v[i] = __shfl_down(a.zzz.v[i], delta);
  }
  return r;
}
__device__ __forceinline__ void msm_finish_hot_body(const TailDesc& d, uint32_t bx);
// The first `hot_blocks` workgroups of the grid are the hot-bucket finishers (msm_finish_hot_body: they and the ordinary workers touch
// disjoint buckets, so one launch serves both — a separate launch was 5 - 9 us on every join's chain even with no hot bucket at all);
// they come first in the grid because a hot bucket's several hundred partials are the longest chain of the launch.
template <bool QUAD, uint32_t FG>
__global__ void __launch_bounds__(256) k_msm_finish(const TailBatch tb, uint32_t hot_blocks) {
  const TailDesc& d = tb.d[blockIdx.y];
  if (blockIdx.x < hot_blocks) {
    msm_finish_hot_body(d, blockIdx.x);
    return;
  }
  const uint32_t bx = blockIdx.x - hot_blocks;
  const uint32_t nb = d.nb;
  constexpr uint32_t LW = QUAD ? 2 : 0;  // log2(lanes per worker)
  const uint32_t worker = (bx * blockDim.x + threadIdx.x) >> LW;
  const uint32_t b = worker / FG, l = worker % FG;
  if ((bx * blockDim.x >> LW) / FG >= nb) return;  // whole workgroup beyond this MSM's buckets
  xyzz29 acc = xyzz29_identity();
  bool hot = false;
  if (b < nb) {
    uint32_t cnt = d.np1[b], s = d.toff1[b];
    hot = cnt > d.hot_min;  // left to k_msm_finish_hot (FG workers would each chain cnt / FG additions)
    if (hot) cnt = 0;
    for (uint32_t k = l; k < cnt; k += FG) {
      xyzz29 p = part_load(d.part1 + (size_t)(s + k) * PART_BYTES);
      if (QUAD) acc = xyzz29_add_quad(acc, p);
      else xyzz29_add(acc, p);
    }
  }
  for (uint32_t dd = FG / 2; dd > 0; dd >>= 1) {
    xyzz29 o = shfl_down_xyzz(acc, dd << LW);  // the same role dd workers up; a bucket spans FG << LW aligned lanes
    if (l < dd) {
      if (QUAD) acc = xyzz29_add_quad(acc, o);
      else xyzz29_add(acc, o);
    }
  }
  if (b < nb && !hot && l == 0 && (!QUAD || (threadIdx.x & 3u) == 0)) part_store(d.dense + (size_t)b * PART_BYTES, acc);
}

// block-wide tree sum of up to 256 XYZZ values held in LDS, one quad per pair
__device__ __forceinline__ void block_tree_sum(xyzz29* lds, uint32_t count_pow2) {
  const uint32_t quad = threadIdx.x >> 2, nquads = blockDim.x >> 2;
  for (uint32_t s = count_pow2 >> 1; s > 0; s >>= 1) {
    for (uint32_t g = quad; g < s; g += nquads) {  // writes go to [0, s), the other operand comes from [s, 2s)
      xyzz29 r = xyzz29_add_quad(lds[g], lds[g + s]);
      if ((threadIdx.x & 3u) == 0) lds[g] = r;
    }
    __syncthreads();
  }
}

// hot buckets (the 0 / 1 buckets of sparse witness columns: hundreds of folded partials): a whole workgroup of
// 64 quads per bucket — each quad chains cnt / 64 additions, then a 6-level tree — instead of 8 workers chaining
// cnt / 8.  A workgroup scans 256 buckets for hot ones (none in the uniform case: it returns after one load).
__device__ __forceinline__ void msm_finish_hot_body(const TailDesc& d, uint32_t bx) {
  __shared__ uint32_t hot_list[256], nhot;
  const uint32_t tid = threadIdx.x, quad = tid >> 2;
  const uint32_t b = bx * 256 + tid;
  if (tid == 0) nhot = 0;
  __syncthreads();
  if (b < d.nb && d.np1[b] > d.hot_min) hot_list[atomicAdd(&nhot, 1u)] = b;
  __syncthreads();
  const uint32_t n_hot = nhot;
  xyzz29* lds = reinterpret_cast<xyzz29*>(h2_msm_smem);
  for (uint32_t h = 0; h < n_hot; h++) {
    const uint32_t hb = hot_list[h];
    const uint32_t cnt = d.np1[hb], s = d.toff1[hb];
    xyzz29 acc = xyzz29_identity();
    for (uint32_t k = quad; k < cnt; k += 64) acc = xyzz29_add_quad(acc, part_load(d.part1 + (size_t)(s + k) * PART_BYTES));
    if ((tid & 3u) == 0) lds[quad] = acc;
    __syncthreads();
    block_tree_sum(lds, 64);
    if (tid == 0) part_store(d.dense + (size_t)hb * PART_BYTES, lds[0]);
    __syncthreads();
  }
}

// wide windows: segment s = buckets [sL, (s+1)L), L = 2^seg_log <= 8: P_s = their sum, V_s = sum_j (L - 1 - j) B_(sL+j) by a
// running sum (V += T; T += B_j) — 2 (L - 1) dependent additions, one QUAD per segment (a lane per segment took 113 us at
// c = 20: the chain's latency, not its throughput, is what a join waits for).  Bucket b sits at the permuted position
// (b & 511) << lb | b >> 9 of the dense array (see WIDE_BIN_BITS).
__global__ void __launch_bounds__(256) k_msm_seg(const TailBatch tb) {
  const TailDesc& d = tb.d[blockIdx.y];
  if (!d.seg_log) return;
  const uint32_t L = 1u << d.seg_log, s = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (s >= (d.nb >> d.seg_log)) return;  // whole quads: the bound is a multiple of 64
  const uint32_t lb = MAT_LOG + d.seg_log - WIDE_BIN_BITS;  // log2(buckets per bin)
  xyzz29 T = xyzz29_identity(), V = xyzz29_identity();
  for (uint32_t j = 0; j < L; j++) {
    const uint32_t b = s * L + j, bp = ((b & ((1u << WIDE_BIN_BITS) - 1)) << lb) | (b >> WIDE_BIN_BITS);
    V = xyzz29_add_quad(V, T);
    T = xyzz29_add_quad(T, part_load(d.dense + (size_t)bp * PART_BYTES));
  }
  if ((threadIdx.x & 3u) == 0) {
    part_store(d.dense2 + (size_t)s * PART_BYTES, T);
    part_store(d.vsum + (size_t)s * PART_BYTES, V);
  }
}

// bucket matrix B[hi][lo] (b = hi*Nl + lo): blocks 0..Nh-1 produce row sums, blocks Nh..Nh+Nl-1 column sums;
// wide windows: blocks Nh+Nl .. 2Nh+Nl-1 the row sums of the segments' local weighted sums (vsum)
__global__ void __launch_bounds__(TAIL_THREADS) k_msm_rowcol(const TailBatch tb) {
  const TailDesc& d = tb.d[blockIdx.y];
  xyzz29* lds = reinterpret_cast<xyzz29*>(h2_msm_smem);
  const uint32_t logNl = d.logNl, Nh = 1u << d.logNh, Nl = 1u << logNl;
  const uint32_t tid = threadIdx.x, blk = blockIdx.x;
  if (blk >= Nh + Nl + (d.seg_log ? Nh : 0u)) return;
  if (tid < 256) {
    xyzz29 v = xyzz29_identity();
    if (blk < Nh) {
      if (tid < Nl) v = part_load(d.mat + (size_t)((blk << logNl) + tid) * PART_BYTES);
    } else if (blk < Nh + Nl) {
      if (tid < Nh) v = part_load(d.mat + (size_t)((tid << logNl) + (blk - Nh)) * PART_BYTES);
    } else {
      if (tid < Nl) v = part_load(d.vsum + (size_t)(((blk - Nh - Nl) << logNl) + tid) * PART_BYTES);
    }
    lds[tid] = v;
  }
  __syncthreads();
  block_tree_sum(lds, max(Nh, Nl));
  if (tid == 0) part_store(d.rc + (size_t)blk * PART_BYTES, lds[0]);
}

// bit-decomposed weights: block beta < logNh sums rows with bit beta of hi set; block logNh + beta sums
// columns with bit beta of (lo+1) set (beta <= logNl).
__global__ void __launch_bounds__(TAIL_THREADS) k_msm_weighted(const TailBatch tb) {
  const TailDesc& d = tb.d[blockIdx.y];
  xyzz29* lds = reinterpret_cast<xyzz29*>(h2_msm_smem);
  const uint32_t logNh = d.logNh, logNl = d.logNl, Nh = 1u << logNh, Nl = 1u << logNl;
  const uint32_t tid = threadIdx.x, blk = blockIdx.x;
  const uint32_t terms = logNh + logNl + 1;
  if (blk >= terms + (d.seg_log ? 1u : 0u)) return;
  if (tid < 256) {
    xyzz29 v = xyzz29_identity();
    if (blk < logNh) {
      if (tid < Nh && ((tid >> blk) & 1u)) v = part_load(d.rc + (size_t)tid * PART_BYTES);
    } else if (blk < terms) {
      uint32_t beta = blk - logNh;
      if (tid < Nl && (((tid + 1) >> beta) & 1u)) v = part_load(d.rc + (size_t)(Nh + tid) * PART_BYTES);
    } else {  // wide windows: the plain sum of the local weighted sums' row sums
      if (tid < Nh) v = part_load(d.rc + (size_t)(Nh + Nl + tid) * PART_BYTES);
    }
    lds[tid] = v;
  }
  __syncthreads();
  block_tree_sum(lds, max(Nh, Nl));
  if (tid == 0) part_store(d.g + (size_t)blk * PART_BYTES, lds[0]);
}

// result = sum_beta 2^(beta + logNl) G_row[beta] + sum_beta 2^beta G_col[beta], returned as a Jacobian
// point in the ABI's Montgomery-2^256 form: (X*ZZ, Y*ZZZ, ZZ) since ZZ^3 = ZZZ^2; identity = (0, R, 0).
// The order of additions inside a bucket follows LDS-atomic arrival in the partition, so the projective
// representative (not the group element) can differ between two runs on the same input: callers compare
// or hash after h2mi_g1_batch_normalize, exactly as the reference's callers do with best_multiexp's result;
// h2mi_msm_set_canonical(1) trades one field inversion per MSM for reproducible bits.
// One workgroup of 32 quads per MSM: quad g doubles term g its 2^shift times, then a 5-level tree.
__global__ void __launch_bounds__(128) k_msm_final(const TailBatch tb) {
  const TailDesc& d = tb.d[blockIdx.x];
  xyzz29* lds = reinterpret_cast<xyzz29*>(h2_msm_smem);
  const uint32_t tid = threadIdx.x, quad = tid >> 2;
  const uint32_t logNh = d.logNh, logNl = d.logNl;
  const uint32_t terms = logNh + logNl + 1;  // <= 17
  xyzz29 v = xyzz29_identity();
  if (quad < terms) {
    v = part_load(d.g + (size_t)quad * PART_BYTES);
    uint32_t shift = quad < logNh ? quad + logNl : quad - logNh;
    for (uint32_t k = 0; k < shift; k++) v = xyzz29_dbl_quad(v);
  }
  if ((tid & 3u) == 0) lds[quad] = v;
  __syncthreads();
  block_tree_sum(lds, 32);
  if (d.seg_log) {  // wide windows: L * (matrix result) - sum of the segments' local weighted sums
    if (quad == 0) {
      xyzz29 r = lds[0];
      for (uint32_t k = 0; k < d.seg_log; k++) r = xyzz29_dbl_quad(r);
      xyzz29 y = part_load(d.g + (size_t)terms * PART_BYTES);
      y.y = f29_normalize(f29_sub(f29_zero(), y.y, Fq29::K4));  // -Y as 4p - Y (Y < 4p)
      r = xyzz29_add_quad(r, y);
      if (tid == 0) lds[0] = r;
    }
    __syncthreads();
  }
  if (tid == 0) {
    xyzz29 r = lds[0];
    jac j;
    if (xyzz29_is_identity(r)) {
      j.x = fe_zero(); j.y = fe_one<Fq>(); j.z = fe_zero();
    } else if (d.canonical) {  // h2mi_msm_set_canonical(1): the representative with Z = 1
      f29 ax, ay;
      xyzz29_to_affine(r, ax, ay);
      f29_to_mont256<Fq29>(ax, j.x.v);
      f29_to_mont256<Fq29>(ay, j.y.v);
      j.z = fe_one<Fq>();
    } else {
      f29_to_mont256<Fq29>(f29_mul<Fq29>(r.x, r.zz), j.x.v);
      f29_to_mont256<Fq29>(f29_mul<Fq29>(r.y, r.zzz), j.y.v);
      f29_to_mont256<Fq29>(r.zz, j.z.v);
    }
    jac_store(d.out, j);
    if (d.stats) d.stats[0] = d.off[d.nb];
  }
}

// ---- small base sets: n <= SMALL_MAX_N (round 4) --------------------------------------------------------------------
// The pipeline above is built for throughput: nine partition kernels, an accumulation over >= 2^12 buckets and a
// five-kernel bucket reduction whose chain of ~40 dependent point operations is the same at every size — 217 us for a
// 256-point MSM (profiles/r03_op_bench.json), six times per proof of the reference's own example (k = 5 .. 8).  A small
// MSM is pure latency: what it costs is the NUMBER OF DEPENDENT POINT OPERATIONS (~3.3 us each for a quad of lanes, measured),
// so this path has no buckets at all:
//   * a second table holds every multiple the signed digits can ask for, M[j][w][i] = j 2^(cs w) P_i, j = 1 .. 2^(cs-1)
//     (cs = 5: 16 x 51 x n points — 13 MB at n = 256, 214 MB at n = 4096, of 288 GB): the MSM is then the plain sum of the
//     <= n W table points its digits select — no bucket weights, no doublings, ceil(log2(n W)) tree levels and nothing else;
//   * k_msm_small_digits (on the caller's stream: the only reader of the scalars): one thread per scalar, Montgomery ->
//     canonical, W signed digits as bytes;
//   * k_msm_small_accum: one lane per r cells (point i, window w) of the digit array — gather, mixed additions — then a quad tree
//     over the workgroup's 128 gathered sums in LDS -> one partial sum per workgroup; the workgroup that arrives last sums the
//     <= 256 partial sums of its MSM the same way and writes the Jacobian result.
// The second kernel is deferred and batched over the MSMs queued since the last join like the bucket reductions above
// (blockIdx.y = MSM): a prover phase of four commitments is four digit launches + one launch.  An earlier form of this path
// (4 .. 16 LDS buckets, per-(bucket, slice) workgroups, bit-sliced weights) needed 21 levels at n = 256 where this one needs 14:
// 131 us of device time against the general pipeline's 298 (profiles/r04_msm_small_sweep.txt, first block).
// the widest base set that takes this path.  Measured (profiles/r04_msm_small_sweep.txt; latency / four MSMs + one join / back to
// back, us, against the general pipeline): 2^8 86 / 138 / 31 vs 229 / 367 / 52; 2^12 132 / 267 / 63 vs 266 / 542 / 83; 2^14 173 / 445 /
// 106 vs 278 / 582 / 98; 2^15 227 / 633 / 150 vs 327 / 727 / 121; 2^16 366 / 1044 / 263 vs 375 / 823 / 169 — from 2^16 the 2.5x more
// mixed additions of the narrow windows (43 x n against 17 x n) cost more than the short chain saves.
constexpr size_t SMALL_MAX_N = (size_t)1 << 14;
constexpr uint32_t SMALL_THREADS = 256, SMALL_PARTS_MAX = 512, SMALL_BATCH = 8, SMALL_C_MAX = 7;
// a workgroup's 256 lanes are 64 quads, and a tree level over N values is N / 2 quad operations: 128 values per workgroup keep
// every level at ONE operation per quad (with 256 the first level ran two in sequence); the upper 128 lanes idle through the
// gather — free, on a path that is bound by its chain and not by throughput
constexpr uint32_t SMALL_CELLS = 128;  // beyond 128 x 256 cells every lane gathers (lanes = 256): the madd chains are what counts there
struct SmallDesc {
  const uint8_t* dig;    // [W][n_reg]: sign << 7 | magnitude (0 = no entry)
  const uint8_t* table;  // [2^(c-1)][W][n_reg] affine points j 2^(c w) P_i (canonical Montgomery-2^261 words)
  uint8_t* part;         // [G] XYZZ partial sums
  uint32_t* cnt;         // [G] entries each workgroup added (statistics); cnt[G_max] = arrival counter of the workgroups
  uint32_t ticket_at;    // index of that counter
  uint8_t* out;
  uint64_t* stats;
  uint32_t n, n_reg, W, c, r, lanes, G, canonical;  // r cells per gathering lane, G = ceil(n W / (lanes r)) workgroups
};
struct SmallBatch {
  SmallDesc d[SMALL_BATCH];
};

// plane j (multiples j Q of the points Q = 2^(c w) P_i of plane 1), j = 2 .. NB: one thread per point, a chain of mixed
// additions with one conversion back to affine per multiple
__global__ void __launch_bounds__(64) k_msm_small_multiples(uint8_t* table, size_t plane /* W * n points */, uint32_t NB) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= plane) return;
  const affine p = affine_load(table + t * 64);
  if (affine_is_identity(p)) {
    for (uint32_t j = 2; j <= NB; j++) affine_store(table + ((size_t)(j - 1) * plane + t) * 64, p);
    return;
  }
  const f29 x = f29_unpack(p.x.v), y = f29_unpack(p.y.v);
  xyzz29 acc = xyzz29_dbl_affine(x, y);
  for (uint32_t j = 2; j <= NB; j++) {
    if (j > 2) xyzz29_madd(acc, x, y);
    f29 ax, ay;
    xyzz29_to_affine(acc, ax, ay);
    affine q;
    f29_pack(ax, q.x.v);
    f29_pack(ay, q.y.v);
    affine_store(table + ((size_t)(j - 1) * plane + t) * 64, q);
  }
}

__device__ __forceinline__ void msm_small_digits_body(const fe* scalars, uint32_t n, uint32_t n_reg, uint32_t c, uint32_t W, uint8_t* dig) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const fe s = msm_scalar(scalars, i, n, nullptr);
  __shared__ uint32_t limbs[64][9];  // dynamically indexed limbs live in LDS, not scratch (row 8 = 0: the word above the top)
#pragma unroll
  for (int k = 0; k < 8; k++) limbs[threadIdx.x][k] = s.v[k];
  limbs[threadIdx.x][8] = 0;
  const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < W; w++) {
    const uint32_t bit = w * c, limb = bit >> 5, sh = bit & 31;
    uint32_t raw = 0;
    if (limb < 8) raw = (uint32_t)((((uint64_t)limbs[threadIdx.x][limb + 1] << 32) | limbs[threadIdx.x][limb]) >> sh) & mask;
    const uint32_t d = raw + carry;  // 0 .. 2^c
    uint32_t mag = d, neg = 0;
    carry = 0;
    if (d > half) {  // negative digit d - 2^c, carry 1
      mag = (1u << c) - d;
      neg = 1;
      carry = 1;
    }
    dig[(size_t)w * n_reg + i] = (uint8_t)(mag ? (neg << 7 | mag) : 0u);
  }
}
__global__ void __launch_bounds__(64) k_msm_small_digits(const fe* scalars, uint32_t n, uint32_t n_reg, uint32_t c, uint32_t W, uint8_t* dig) {
  msm_small_digits_body(scalars, n, n_reg, c, W, dig);
}
struct SmallHeadBatch {  // the digit kernels of up to HEAD_BATCH MSMs over one small base set as one launch (blockIdx.y = MSM)
  const fe* scalars[HEAD_BATCH];
  uint8_t* dig[HEAD_BATCH];
};
__global__ void __launch_bounds__(64) k_msm_small_digits_b(const SmallHeadBatch hb, uint32_t n, uint32_t n_reg, uint32_t c, uint32_t W) {
  msm_small_digits_body(hb.scalars[blockIdx.y], n, n_reg, c, W, hb.dig[blockIdx.y]);
}

// segmented tree sums in LDS: nseg segments of len (a power of two) consecutive XYZZ values each; segment k's sum ends
// in lds[k * len].  One quad per pair; a level's pairs are spread over the workgroup's quads.
__device__ __forceinline__ void seg_tree_sum(xyzz29* lds, uint32_t nseg, uint32_t len) {
  const uint32_t quad = threadIdx.x >> 2, nquads = blockDim.x >> 2;
  for (uint32_t h = len >> 1; h > 0; h >>= 1) {
    const uint32_t ops = nseg * h;
    for (uint32_t op = quad; op < ops; op += nquads) {
      const uint32_t at = (op / h) * len + (op % h);
      xyzz29 r = xyzz29_add_quad(lds[at], lds[at + h]);
      if ((threadIdx.x & 3u) == 0) lds[at] = r;
    }
    __syncthreads();
  }
}

// sum of the G partial sums -> Jacobian result (+ statistics); X: LDS for >= SMALL_PARTS_MAX values.  Whole workgroup.
__device__ __forceinline__ void small_finish(const SmallDesc& d, xyzz29* X) {
  const uint32_t tid = threadIdx.x, G = d.G;
  __shared__ uint32_t ins_sh;
  if (tid == 0) ins_sh = 0;
  uint32_t T = 1;
  while (T < G) T <<= 1;
  __syncthreads();
  for (uint32_t t = tid; t < T; t += SMALL_THREADS) {  // G <= SMALL_PARTS_MAX = 512 values, 256 lanes
    X[t] = t < G ? part_load(d.part + (size_t)t * PART_BYTES) : xyzz29_identity();
    if (t < G) atomicAdd(&ins_sh, d.cnt[t]);
  }
  __syncthreads();
  seg_tree_sum(X, 1, T);
  if (tid == 0) {
    xyzz29 r = X[0];
    jac j;
    if (xyzz29_is_identity(r)) {
      j.x = fe_zero(); j.y = fe_one<Fq>(); j.z = fe_zero();
    } else if (d.canonical) {
      f29 ax, ay;
      xyzz29_to_affine(r, ax, ay);
      f29_to_mont256<Fq29>(ax, j.x.v);
      f29_to_mont256<Fq29>(ay, j.y.v);
      j.z = fe_one<Fq>();
    } else {
      f29_to_mont256<Fq29>(f29_mul<Fq29>(r.x, r.zz), j.x.v);
      f29_to_mont256<Fq29>(f29_mul<Fq29>(r.y, r.zzz), j.y.v);
      f29_to_mont256<Fq29>(r.zz, j.z.v);
    }
    jac_store(d.out, j);
    if (d.stats) d.stats[0] = ins_sh;
  }
}

// One launch per batch: every workgroup leaves its partial sum, and the workgroup that ARRIVES LAST (a device-scope counter;
// nobody waits for anybody, so there is nothing to deadlock) sums the partial sums of its MSM and writes the result — a second
// launch cost ~8 us of the ~85 (launch + the first touch of the partial sums through a cold L2).  Release / acquire at agent
// scope (__threadfence = the partial sums are written back before the counter moves, and re-read from memory after it):
// the workgroups of one launch are spread over the eight XCDs, whose L2s are not coherent for plain accesses.
__global__ void __launch_bounds__(SMALL_THREADS) k_msm_small_accum(const SmallBatch sb) {
  const SmallDesc& d = sb.d[blockIdx.y];
  if (blockIdx.x >= d.G) return;  // a batch is launched over its largest member
  const uint32_t tid = threadIdx.x;
  const uint32_t cells = d.n * d.W;           // cell = w * n + i (the digit array has row length n_reg)
  const size_t plane = (size_t)d.W * d.n_reg;  // points per multiple
  xyzz29* tree = reinterpret_cast<xyzz29*>(h2_msm_smem);  // SMALL_PARTS_MAX values
  xyzz29 acc = xyzz29_identity();
  uint32_t mine = 0;
  const uint32_t lanes = d.lanes;
  const uint32_t base = blockIdx.x * lanes * d.r;
  for (uint32_t k = 0; k < d.r && tid < lanes; k++) {
    const uint32_t cell = base + k * lanes + tid;
    if (cell >= cells) break;
    const uint32_t w = cell / d.n, at = w * d.n_reg + (cell - w * d.n);
    const uint32_t dg = d.dig[at];
    if (!dg) continue;
    const affine p = affine_load(d.table + ((size_t)((dg & 0x7fu) - 1) * plane + at) * 64);
    if (affine_is_identity(p)) continue;
    mine++;
    f29 x2 = f29_unpack(p.x.v), y2 = f29_unpack(p.y.v);
    if (dg >> 7) y2 = f29_sub(f29_zero(), y2, Fq29::K2);  // 2p - y (lazy)
    xyzz29_madd(acc, x2, y2);
  }
  if (tid < lanes) tree[tid] = acc;
  __shared__ uint32_t cnt_sh, last_sh;
  if (tid == 0) cnt_sh = 0;
  __syncthreads();
  if (mine) atomicAdd(&cnt_sh, mine);
  uint32_t live = min(cells - base, lanes), T = 1;  // lanes that had a cell at all (k = 0)
  while (T < live) T <<= 1;
  seg_tree_sum(tree, 1, T);  // (its first barrier also orders the counter)
  if (tid == 0) {
    part_store(d.part + (size_t)blockIdx.x * PART_BYTES, tree[0]);
    d.cnt[blockIdx.x] = cnt_sh;
    __threadfence();  // release: this workgroup's partial sum and count are visible device-wide before its arrival is
    const uint32_t ticket = atomicAdd(&d.cnt[d.ticket_at], 1u);
    last_sh = ticket == d.G - 1 ? 1u : 0u;
    if (last_sh) {
      d.cnt[d.ticket_at] = 0;  // ready for the slot's next MSM (stream order: no workgroup of this MSM touches it again)
      __threadfence();          // acquire: the other workgroups' partial sums are read from memory, not from a stale line
    }
  }
  __syncthreads();
  if (!last_sh) return;  // workgroup-uniform
  __threadfence();       // every lane that is about to read the partial sums orders its loads behind the counter
  small_finish(d, tree);
}

// ---- host -----------------------------------------------------------------------------------------
static uint32_t pick_window(size_t n) {
  const char* ev = getenv("H2MI_MSM_C");
  if (ev) {
    int c = atoi(ev);
    if (c >= 11 && c <= (int)WIDE_MAX_C) return (uint32_t)c;  // below 11 the scatter tile (1024 x W pairs) outgrows LDS
  }
  uint32_t lg = 0;
  while (((size_t)1 << lg) < n) lg++;
  // Larger windows mean fewer point additions (n * ceil(255/c)); the bucket phase is latency-bound and
  // nearly independent of the bucket count.  Only windows whose TOP window still holds many scalar bits
  // are used: c = 12 or 14 leave 2 bits there, i.e. four buckets that each receive n/4 points.
  // c = 17 (15 windows, the widest the bucket-matrix kernels take): 6 % fewer additions for twice the buckets to
  // reduce.  Re-measured at the end of round 2 inside whole proofs (the reductions are cheaper than when this table was
  // first drawn up): create_proof 2^20 -1..2 % (the replay step unchanged), 2^21 -5 %, 2^22 -3 % — the default from 2^21.
  // Round 3: wide windows (18 .. 20 bits: 2^17 .. 2^19 buckets) measured — profiles/r03_msm_sweep.txt.  c = 18 leaves a 2-bit top
  // window (four buckets take a fifteenth of the MSM: 2x slower); c = 19 / 20 cut the accumulation by 10 - 15 % at every size,
  // but their bucket reduction is throughput-bound (2^19 buckets x ~3 additions: fold, finish, segment sums) and costs more
  // than the accumulation saves at 2^20 (replay step 20.0 -> 21.4 ms, create_proof 14.1 -> 14.8 ms) and breaks even at 2^21; from
  // 2^22 (104 entries per bucket) c = 20 wins: accumulation 5.04 -> 4.39 ms, range proof at DEGREE 22 73.7 -> 72.3 ms.
  // Smaller sizes, same sweep (`tools/msm_sweep.py`, back-to-back / latency in us): 2^17: c = 13: 307 / 517, 15: 287 / 494,
  // 16: 286 / 476; 2^18: 15: 475 / 706, 16: 458 / 666; 2^19: 15: 832 / 1112, 16: 809 / 1058; 2^15: 13: 180 / 334, 16: 164 /
  // 324; 2^14: 13: 149 / 296, 15: 135 / 282, 16: 153 / 309; 2^10: 13 is best.  (c = 14 leaves a 3-bit top window: 50 % slower.)
  // Round 3, after the shared-reduction Y3 (f29_mul2) and the cheaper reductions: c = 17 re-measured against 16, alternating on one
  // box: 2^20: create_proof 13.2 - 14.4 -> 12.8 - 13.0 ms (minimum of 12), replay step 19.2 - 19.9 -> 18.8 - 19.1 ms; 2^19: neutral
  // (MSM 764 -> 732 us back-to-back, proofs 8.0 - 8.2 either way); 2^18 and 2^17: 16 stays (434 -> 455, 249 -> 266 us) — 17 from 2^20.
  // 2^15 / 2^16 re-measured inside whole proofs (C++ host, alternating, late round 3): 15-bit windows 3.04 / 3.24 ms against 3.10 / 3.37 with
  // 16 (the halo2_lib shape at 2^16: 3.32 against 3.58) — half the buckets for the latency-bound reductions outweigh one more window;
  // equal at 2^17 and 2^18: 16 from 2^17.
  int c = lg >= 22 ? 20 : lg >= 20 ? 17 : lg >= 17 ? 16 : lg >= 11 ? 15 : 13;
  return (uint32_t)c;
}

// small base sets: window width, gathering lanes per workgroup and cells per lane (see "small base sets").  Up to 128 x 256
// cells: 128 lanes, one cell each; beyond: 256 lanes and r = ceil(n W / (256 x 512)) cells each (<= 512 workgroups = two
// wavefronts per SIMD: a lone wavefront issues a multiply-add every 8 cycles, two share the unit at ~4.8).  The window is the
// widest whose table 2^(c-1) x W x n x 64 B stays within SMALL_TABLE_BUDGET.  Returns false when the set does not take the path.
// First guesses by level count, then measured (tools/msm_small_sweep.sh -> profiles/r04_msm_small_sweep.txt).
constexpr size_t SMALL_TABLE_BUDGET = (size_t)2 << 30;  // per base set: c = 7 up to 2^13 points (1.2 GB), c = 6 at 2^14 (1.4 GB)
static bool small_geometry(size_t n, uint32_t* c, uint32_t* lanes, uint32_t* r) {
  size_t max_n = SMALL_MAX_N;
  if (const char* ev = ab_env("H2MI_MSM_SMALL_MAX_LOG"))
    max_n = std::min<size_t>(SMALL_MAX_N, (size_t)1 << std::max(0, atoi(ev)));
  if (n > max_n) return false;
  if (getenv("H2MI_MSM_C")) return false;  // a forced window width means the general pipeline with that width (forced-path parity tests)
  uint32_t cc = SMALL_C_MAX;
  while (cc > 2 && ((size_t)1 << (cc - 1)) * ((255 + cc - 1) / cc) * n * 64 > SMALL_TABLE_BUDGET) cc--;
  if (const char* ev = ab_env("H2MI_MSM_SMALL_C"))
    if (atoi(ev) >= 2 && atoi(ev) <= (int)SMALL_C_MAX) cc = (uint32_t)atoi(ev);
  const size_t cells = n * ((255 + cc - 1) / cc);
  uint32_t ll = SMALL_CELLS, rr = 1;
  if (cells > (size_t)SMALL_CELLS * 256) {
    ll = SMALL_THREADS;
    rr = (uint32_t)((cells + (size_t)SMALL_THREADS * SMALL_PARTS_MAX - 1) / ((size_t)SMALL_THREADS * SMALL_PARTS_MAX));
  }
  if (const char* ev = ab_env("H2MI_MSM_SMALL_R"))
    if (atoi(ev) >= (int)rr && atoi(ev) <= 256) rr = (uint32_t)atoi(ev), ll = SMALL_THREADS;
  *c = cc;
  *lanes = ll;
  *r = rr;
  return true;
}

// H2MI_MSM_S0 fixes the chunk length (tuning / tests); 0 = chosen on the device from the entry count
static uint32_t chunk_override() {
  const char* ev = getenv("H2MI_MSM_S0");
  return (ev && atoi(ev) >= 1 && atoi(ev) <= (int)S0_MAX) ? (uint32_t)atoi(ev) : 0u;
}

static void free_bases(Bases* B) {
  H2_IGNORE(hipFree(B->table));
  H2_IGNORE(hipFree(B->stable));
  H2_IGNORE(hipFree(B->host_stage));
  for (Slot& S : B->slot) {
    H2_IGNORE(hipFree(S.vals[0])); H2_IGNORE(hipFree(S.vals[1]));
    H2_IGNORE(hipFree(S.bkeys)); H2_IGNORE(hipFree(S.bincnt)); H2_IGNORE(hipFree(S.binbase)); H2_IGNORE(hipFree(S.binseg)); H2_IGNORE(hipFree(S.tile_live));
    H2_IGNORE(hipFree(S.off)); H2_IGNORE(hipFree(S.s0_dev));
    if (S.pa_spill) { H2_IGNORE(hipFree(S.pa_spill)); H2_IGNORE(hipFree(S.pa_tot)); }
    for (int i = 0; i < 2; i++) { H2_IGNORE(hipFree(S.np[i])); H2_IGNORE(hipFree(S.toff[i])); }
    H2_IGNORE(hipFree(S.dense)); H2_IGNORE(hipFree(S.dense2)); H2_IGNORE(hipFree(S.vsum)); H2_IGNORE(hipFree(S.tseg[0])); H2_IGNORE(hipFree(S.tseg[1]));
    H2_IGNORE(hipFree(S.part[0])); H2_IGNORE(hipFree(S.part[1])); H2_IGNORE(hipFree(S.rc)); H2_IGNORE(hipFree(S.g)); H2_IGNORE(hipFree(S.stats)); H2_IGNORE(hipFree(S.shift));
    H2_IGNORE(hipFree(S.sdig)); H2_IGNORE(hipFree(S.spart)); H2_IGNORE(hipFree(S.scnt));
    if (S.input_ready) H2_IGNORE(hipEventDestroy(S.input_ready));
    if (S.head_done) H2_IGNORE(hipEventDestroy(S.head_done));
    if (S.accum_done) H2_IGNORE(hipEventDestroy(S.accum_done));
    if (S.tail_done) H2_IGNORE(hipEventDestroy(S.tail_done));
  }
  delete B;
}

#define H2_ALLOC(ptr, bytes)                                   \
  do {                                                         \
    hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));        \
    if (e_ != hipSuccess) {                                    \
      note_hip_error(e_, __FILE__, __LINE__);                  \
      free_bases(B);                                           \
      return e_ == hipErrorOutOfMemory ? H2MI_ENOMEM : H2MI_EHIP; \
    }                                                          \
  } while (0)

static int msm_dev(Bases* B, const void* d_scalars, size_t n, void* d_out, hipStream_t s, bool inorder = false, bool general = false);

static int register_dev(const void* d_bases, size_t n, uint64_t* handle_out, hipStream_t s, bool allow_small = true) {
  if (n == 0 || n > ((size_t)1 << 26)) return H2MI_ERANGE;
  Bases* B = new Bases();
  B->n = n;
  B->dev = ctx().cur;
  B->stride = n + 1;
  B->c = pick_window(n);
  B->W = (255 + B->c - 1) / B->c;
  B->nb = 1u << (B->c - 1);
  B->seg_log = B->c - 1 > MAT_LOG ? B->c - 1 - MAT_LOG : 0;
  B->logNl = (B->c - 1 - B->seg_log + 1) / 2;
  B->logNh = (B->c - 1 - B->seg_log) - B->logNl;
  if ((uint64_t)B->stride * B->W >= (1ull << 31)) { delete B; return H2MI_ERANGE; }
  const size_t nW = B->stride * B->W;
  // partial sums of one accumulation: one per chunk (at most whole rounds of the resident grid) + one per bucket
  B->max_tasks0 = std::min(accum_rounds((uint32_t)nW) * ACCUM_RESIDENT_CHUNKS, (uint32_t)nW) + B->nb + 1;
  B->max_tasks1 = B->max_tasks0 / S1 + B->nb;
  H2_ALLOC(B->table, nW * 64);
  B->lb = B->c - 1 > 9 ? B->c - 1 - 9 : 0;
  B->nbins = B->nb >> B->lb;  // <= NBINS_MAX
  const size_t ntiles_max = (B->stride + P1_TS - 1) / P1_TS;
  const size_t bin_cells = (size_t)B->nbins * ntiles_max + 1;
  if (bin_cells >= ((size_t)1 << 31)) { free_bases(B); return H2MI_ERANGE; }
  // the own scans (k_scan_seg) take 16-byte vectors: the [bin][tile] matrix has 512 rows and the task arrays
  // 2^(c-1) >= 1024 entries for every window width pick_window() can return
  if ((B->nbins & 3u) || B->nb < 4 || (B->nb > SCAN_SEG_TASKS && !B->seg_log) || B->nbins > NBINS_MAX || (1u << B->lb) > NQW) { free_bases(B); return H2MI_ERANGE; }
  B->nslot = n > ((size_t)1 << 17) ? 4 : NSLOT;
  if (const char* ev = ab_env("H2MI_MSM_SLOTS"))  // A/B knob
    if (atoi(ev) >= 2 && atoi(ev) <= NSLOT) B->nslot = atoi(ev);
  for (int si_ = 0; si_ < B->nslot; si_++) {
    Slot& S = B->slot[si_];
    for (int i = 0; i < 2; i++) H2_ALLOC(S.vals[i], nW * 4);
    H2_ALLOC(S.bkeys, nW * (B->seg_log ? 2 : 1) + 16);
    H2_ALLOC(S.bincnt, bin_cells * 4);
    H2_ALLOC(S.binbase, bin_cells * 4);
    H2_ALLOC(S.binseg, (bin_cells / SCAN_SEG_BINS + 2) * 4);
    H2_ALLOC(S.tile_live, (ntiles_max + 1) * 4);
    H2_ALLOC(S.off, (size_t)(B->nb + 1) * 4);
    H2_ALLOC(S.s0_dev, 4);
    for (int i = 0; i < 2; i++) {
      H2_ALLOC(S.np[i], (size_t)(B->nb + 1) * 4);
      H2_ALLOC(S.toff[i], (size_t)(B->nb + 1) * 4);
    }
    H2_ALLOC(S.dense, (size_t)B->nb * PART_BYTES);
    if (B->seg_log) {
      H2_ALLOC(S.dense2, ((size_t)1 << MAT_LOG) * PART_BYTES);
      H2_ALLOC(S.vsum, ((size_t)1 << MAT_LOG) * PART_BYTES);
      for (int i = 0; i < 2; i++) H2_ALLOC(S.tseg[i], (size_t)(B->nb / SCAN_SEG_BINS + 2) * 4);
    }
    H2_ALLOC(S.part[0], (size_t)B->max_tasks0 * PART_BYTES);
    H2_ALLOC(S.part[1], (size_t)B->max_tasks1 * PART_BYTES);
    H2_ALLOC(S.rc, (size_t)((2u << B->logNh) + (1u << B->logNl)) * PART_BYTES);  // row, column (and wide: vsum row) sums
    H2_ALLOC(S.g, (size_t)64 * PART_BYTES);
    H2_ALLOC(S.stats, 64);
    H2_ALLOC(S.shift, 32);
    if (hipEventCreateWithFlags(&S.input_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&S.head_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&S.accum_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&S.tail_done, hipEventDisableTiming) != hipSuccess) {
      free_bases(B);
      return H2MI_EHIP;
    }
  }
  {
    const bool prof_ = prof_on("k_msm_table_first");
    if (prof_) prof_begin("k_msm_table_first", s);
    hipLaunchKernelGGL(k_msm_table_first, dim3(ceil_div_u32(n, 256)), dim3(256), 0, s, (const uint8_t*)d_bases, B->table, n);
    if (prof_) prof_end(s);
  }
  for (uint32_t w = 1; w < B->W; w++) {
    const bool prof_ = prof_on("k_msm_table_next");
    if (prof_) prof_begin("k_msm_table_next", s);
    hipLaunchKernelGGL(k_msm_table_next, dim3(ceil_div_u32(n, 256)), dim3(256), 0, s, (const uint8_t*)(B->table + (size_t)(w - 1) * B->stride * 64),
                       B->table + (size_t)w * B->stride * 64, n, B->c);
    if (prof_) prof_end(s);
  }
  // (an ad-hoc base set — handle = 0: used once or a few times — keeps the general pipeline: the multiples table costs ~10 ms of
  // dependent conversions to build, which pays for a registered SRS and not for a one-off slice)
  if (allow_small && small_geometry(n, &B->sc, &B->slanes, &B->sr)) {  // second table for the latency path: every multiple a signed digit can select
    B->sW = (255 + B->sc - 1) / B->sc;
    B->sG = ceil_div_u32((uint64_t)n * B->sW, (uint64_t)B->slanes * B->sr);
    const uint32_t NBs = 1u << (B->sc - 1);
    const size_t plane = (size_t)B->sW * n;
    // the latency path is an extra: when its table (up to SMALL_TABLE_BUDGET) or scratch does not fit, the handle keeps the general
    // pipeline instead of failing the registration (round-4 ADVICE)
    bool fits = hipMalloc((void**)&B->stable, (size_t)NBs * plane * 64) == hipSuccess;
    for (int si_ = 0; fits && si_ < B->nslot; si_++) {
      Slot& S = B->slot[si_];
      fits = hipMalloc((void**)&S.sdig, plane) == hipSuccess && hipMalloc((void**)&S.spart, (size_t)B->sG * PART_BYTES) == hipSuccess &&
             hipMalloc((void**)&S.scnt, (size_t)(B->sG + 1) * 4) == hipSuccess;
      if (fits && hipMemsetAsync(S.scnt, 0, (size_t)(B->sG + 1) * 4, s) != hipSuccess) { free_bases(B); return H2MI_EHIP; }  // [sG] = the arrival counter
    }
    if (!fits) {
      (void)hipGetLastError();
      H2_IGNORE(hipStreamSynchronize(s));
      for (int si_ = 0; si_ < B->nslot; si_++) {
        Slot& S = B->slot[si_];
        if (S.sdig) H2_IGNORE(hipFree(S.sdig));
        if (S.spart) H2_IGNORE(hipFree(S.spart));
        if (S.scnt) H2_IGNORE(hipFree(S.scnt));
        S.sdig = nullptr; S.spart = nullptr; S.scnt = nullptr;
      }
      if (B->stable) H2_IGNORE(hipFree(B->stable));
      B->stable = nullptr;
    }
    if (fits) {
    if (hipMemcpyAsync(B->stable, B->table, n * 64, hipMemcpyDeviceToDevice, s) != hipSuccess) { free_bases(B); return H2MI_EHIP; }
    for (uint32_t w = 1; w < B->sW; w++)
      hipLaunchKernelGGL(k_msm_table_next, dim3(ceil_div_u32(n, 64)), dim3(64), 0, s, (const uint8_t*)(B->stable + (size_t)(w - 1) * n * 64),
                         B->stable + (size_t)w * n * 64, n, B->sc);
    hipLaunchKernelGGL(k_msm_small_multiples, dim3(ceil_div_u32(plane, 64)), dim3(64), 0, s, B->stable, plane, NBs);
    B->small = true;
    }
  }
  // slot n of every window stays the identity until the sum point is known
  for (uint32_t w = 0; w < B->W; w++)
    if (hipMemsetAsync(B->table + ((size_t)w * B->stride + n) * 64, 0, 64, s) != hipSuccess) { free_bases(B); return H2MI_EHIP; }
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { free_bases(B); return H2MI_EHIP; }
  // B = sum of the bases, by an all-ones MSM through the pipeline just built (the shift is off while has_sum is
  // false), then its own table column.  Small base sets never use the shift (k_msm_pick_shift samples 64 scalars).
  const uint64_t h = g_next_handle++;
  g_bases[h] = B;  // msm_join_all walks the registered handles
  if (n >= SHIFT_MIN_N && !B->small && !ab_env("H2MI_MSM_NO_SHIFT")) {
    fe* ones = nullptr;
    uint8_t* sum = nullptr;
    int rc = H2MI_OK;
    if (hipMalloc((void**)&ones, n * 32) != hipSuccess || hipMalloc((void**)&sum, 96) != hipSuccess) rc = H2MI_ENOMEM;
    if (!rc) {
      hipLaunchKernelGGL(k_msm_fill_one, dim3(ceil_div_u32(n, 256)), dim3(256), 0, s, ones, n);
      rc = msm_dev(B, ones, n, sum, s);
    }
    if (!rc) rc = msm_join_all(s);
    if (!rc) {
      hipLaunchKernelGGL(k_msm_table_sum_point, dim3(1), dim3(64), 0, s, (const uint8_t*)sum, B->table, B->stride, n, B->W, B->c);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = H2MI_EHIP;
    }
    H2_IGNORE(hipFree(ones));
    H2_IGNORE(hipFree(sum));
    if (rc) {
      H2_IGNORE(hipDeviceSynchronize());
      g_bases.erase(h);
      free_bases(B);
      return rc;
    }
    B->has_sum = true;
  }
  *handle_out = h;
  return H2MI_OK;
}

struct Deferred {
  Bases* B;
  Slot* S;
};
// ---- one prover process, several devices (h2mi_init_devices) ---------------------------------------------------
// A sharded handle owns one ordinary registration per device, each over a contiguous slice of the bases — the
// partition best_multiexp applies per CPU thread and the torchrun path applies per rank.  An MSM
// on it launches every slice from the calling thread (scalars: host -> each device, or peer copies from the primary
// device), and the 96-byte partial results are gathered to the primary device and folded there at the next join.
struct Shard {
  uint64_t handle = 0;       // ordinary handle in g_bases
  size_t lo = 0, hi = 0;     // slice [lo, hi) of the registered bases
  uint8_t* stage = nullptr;  // on the shard's device: scalars of its slice (lazy)
  uint8_t* part = nullptr;   // on the shard's device: ring of Jacobian partial results, 96 B each
  hipEvent_t ready = nullptr;
};
constexpr uint32_t SHARD_RING = 16;  // sharded MSMs that may be queued between two joins
struct Sharded {
  size_t n = 0;
  std::vector<Shard> shard;
  uint8_t* gather = nullptr;  // on the primary device: [ring][shard] partial results
  uint32_t next = 0;
};
struct PendingFold {
  Sharded* sh;
  uint32_t ring;
  void* d_out;  // on the primary device
};
static std::map<uint64_t, Sharded*> g_sharded;
static std::vector<PendingFold> g_pending_folds;
static int fold_sharded(hipStream_t s0);
static std::vector<Deferred> g_deferred;
static int flush_tails();
static int launch_tails(const TailBatch& tb, uint32_t count, uint32_t max_tasks1, uint32_t max_nb, uint32_t max_logNh, uint32_t max_logNl,
                        uint32_t max_seg, hipStream_t t);
static TailDesc tail_desc(const Bases* B, const Slot& S);

// One MSM.  On the library's own stream the call is split over internal streams and only queues the
// partition and the accumulation; the latency-bound bucket reduction is deferred to the next join
// (msm_join_all: h2mi_join / h2mi_sync / h2mi_memcpy_d2h), a full batch, or the reuse of the slot.
// On a caller-provided stream everything runs in order on that stream.
static int msm_small(Bases* B, const void* d_scalars, size_t n, void* d_out, hipStream_t s, bool inorder = false);
static bool take_small(const Bases* B, bool general) {
  return B->small && !general && !(B->n > SMALL_STREAM_N && B->since_join >= SMALL_STREAM_AFTER && !ab_env("H2MI_MSM_NO_AUTO_STREAM"));
}
constexpr size_t HEAD_BATCH_MAX_N = (size_t)1 << 17;  // largest base set whose MSMs are partitioned and accumulated as a batch

// `m` <= HEAD_BATCH MSMs of n scalars each over B, results to d_out + 96 j: what m calls of msm_dev would compute, with the partition
// and the accumulation of all m as ONE set of launches (narrow windows, library stream).  Slots, events and the deferred bucket
// reductions are those of m single calls.
static int msm_dev_batch(Bases* B, const void* const* d_scalars, size_t m, size_t n, void* d_out, hipStream_t s, bool inorder) {
  const uint32_t nb = B->nb, W = B->W;
  const bool shifted = B->has_sum && n == B->n;
  const size_t n_eff = n + (shifted ? 1 : 0);
  const uint32_t total = (uint32_t)(n_eff * W);
  // the second partition level and the accumulation share ONE stream here: the separate accumulation stream exists so that the next
  // MSM's partition overlaps this one's accumulation, which a batch does not need — and a stream hop is ~10 us on the phase's critical path
  // `inorder` (the batch is all its phase commits and is read back next): everything, the bucket reductions included, on `s` itself
  hipStream_t hs = inorder ? s : ctx().head_stream, as = hs;
  uint32_t s0_fixed = chunk_override();
  while (s0_fixed && s0_fixed < S0_MAX && ((uint64_t)total + s0_fixed - 1) / s0_fixed + nb > B->max_tasks0) s0_fixed++;
  if (s0_fixed && ((uint64_t)total + s0_fixed - 1) / s0_fixed + nb > B->max_tasks0) return H2MI_ERANGE;
  if ((size_t)P1_TS * W * 6 > 150 * 1024) return H2MI_ERANGE;
  Slot* slots[HEAD_BATCH];
  HeadBatch hb;
  memset(&hb, 0, sizeof(hb));
  for (size_t j = 0; j < m; j++) {
    Slot& S = B->slot[B->next_slot];
    B->last_slot = B->next_slot;
    B->next_slot = (B->next_slot + 1) % B->nslot;
    if (S.tail_deferred) {  // this slot still waits for its reduction (never one of this batch: m <= nslot)
      int rc = flush_tails();
      if (rc) return rc;
    }
    if (S.head_pending) H2_HIP(hipStreamWaitEvent(s, S.head_done, 0));
    if (S.tail_pending && S.last_stream != s) H2_HIP(hipStreamWaitEvent(s, S.tail_done, 0));
    S.last_stream = s;
    slots[j] = &S;
    HeadDesc& d = hb.d[j];
    d.scalars = (const fe*)d_scalars[j];
    d.shift = shifted ? S.shift : nullptr;
    d.bincnt = S.bincnt; d.tile_live = S.tile_live; d.binseg = S.binseg; d.binbase = S.binbase;
    d.vals0 = S.vals[0]; d.vals1 = S.vals[1]; d.bkeys = S.bkeys;
    d.off = S.off; d.np0 = S.np[0]; d.np1 = S.np[1]; d.toff0 = S.toff[0]; d.toff1 = S.toff[1]; d.s0_dev = S.s0_dev;
    d.part0 = S.part[0];
  }
  for (size_t j = m; j < HEAD_BATCH; j++) hb.d[j] = hb.d[0];  // never launched (gridDim.y = m)
  const uint32_t ntiles = ceil_div_u32(n_eff, P1_TS), mm = (uint32_t)m;
  static bool attr_set_dev[16] = {};  // function attributes are per device (every shard of a sharded handle launches these)
  bool& attr_set = attr_set_dev[ctx().cur & 15];
  if (!attr_set) {
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter_b<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter_b<13>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter_b<15>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter_b<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter_b<17>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
#define H2_BIN_COUNT_B(CT) H2_LAUNCH("k_msm_bin_count", k_msm_bin_count_b<CT>, dim3(ntiles, mm), P1_TS, 0, s, hb, n, B->c, W, B->lb, B->nbins, ntiles)
#define H2_BIN_SCATTER_B(CT) \
  H2_LAUNCH("k_msm_bin_scatter", k_msm_bin_scatter_b<CT>, dim3(ntiles, mm), P1_TS, (size_t)P1_TS * W * 6, s, hb, n, B->stride, B->c, W, B->lb, B->nbins, ntiles)
  if (shifted) H2_LAUNCH("k_msm_pick_shift", k_msm_pick_shift_b, dim3(1, mm), 64, 0, s, hb, n);
  switch (B->c) {
    case 13: H2_BIN_COUNT_B(13); break;
    case 15: H2_BIN_COUNT_B(15); break;
    case 16: H2_BIN_COUNT_B(16); break;
    case 17: H2_BIN_COUNT_B(17); break;
    default: H2_BIN_COUNT_B(0); break;
  }
  const uint32_t cells = B->nbins * ntiles;
  const uint32_t nseg = ceil_div_u32(cells, SCAN_SEG_BINS);
  if (nseg > 1) H2_LAUNCH("k_scan_segsum", k_scan_segsum_bins_b, dim3(nseg, mm), 1024, 0, s, hb, cells);
  H2_LAUNCH("k_scan_seg_bins", k_scan_seg_bins_b, dim3(nseg, mm), 1024, 0, s, hb, cells);
  switch (B->c) {
    case 13: H2_BIN_SCATTER_B(13); break;
    case 15: H2_BIN_SCATTER_B(15); break;
    case 16: H2_BIN_SCATTER_B(16); break;
    case 17: H2_BIN_SCATTER_B(17); break;
    default: H2_BIN_SCATTER_B(0); break;
  }
#undef H2_BIN_COUNT_B
#undef H2_BIN_SCATTER_B
  // one event for the batch's inputs; the slots' own events for what later calls wait on per slot
  if (hs != s) {
    H2_HIP(hipEventRecord(slots[0]->input_ready, s));
    H2_HIP(hipStreamWaitEvent(hs, slots[0]->input_ready, 0));
  }
  for (size_t j = 0; j < m; j++) {
    Slot& S = *slots[j];
    if (S.accum_pending) H2_HIP(hipStreamWaitEvent(hs, S.accum_done, 0));
    if (S.tail_pending) H2_HIP(hipStreamWaitEvent(hs, S.tail_done, 0));
    S.tail_pending = S.accum_pending = S.head_pending = false;
  }
  H2_LAUNCH("k_msm_bin_sort", k_msm_bin_sort_b, dim3(B->nbins, mm), P2_THREADS, 0, hs, hb, ntiles, B->nbins, B->lb, s0_fixed, nb);
  H2_LAUNCH("k_scan_seg_tasks", k_scan_seg_tasks_b, dim3(1, 2, mm), 1024, 0, hs, hb, nb);
  const uint32_t chunks0 = s0_fixed ? (uint32_t)(((uint64_t)total + s0_fixed - 1) / s0_fixed) : std::min(accum_rounds(total) * ACCUM_RESIDENT_CHUNKS, total);
  const uint32_t tasks0 = chunks0 + nb;
  for (size_t j = 0; j < m; j++) {
    H2_HIP(hipEventRecord(slots[j]->head_done, hs));
    slots[j]->head_pending = true;
    slots[j]->head_ever = true;
  }
  if (as != hs) H2_HIP(hipStreamWaitEvent(as, slots[m - 1]->head_done, 0));
  static const size_t accum_lds = ab_env("H2MI_ACCUM_LDS") ? (size_t)atoi(ab_env("H2MI_ACCUM_LDS")) : 56000;
  H2_LAUNCH("k_msm_accum", k_msm_accum_b, dim3(ceil_div_u32(chunks0, 256), mm), 256, accum_lds, as, hb, nb, (const uint8_t*)B->table);
  for (size_t j = 0; j < m; j++) {
    Slot& S = *slots[j];
    S.d_out = (char*)d_out + 96 * j;
    S.tasks1 = tasks0 / S1 + nb;
    H2_HIP(hipEventRecord(S.accum_done, as));
    S.accum_pending = true;
    S.accum_ever = true;
    if (!inorder) {
      S.tail_deferred = true;
      g_deferred.push_back({B, &S});
    }
  }
  if (inorder) {  // the batch's own reductions at once, behind its accumulation on the same stream
    TailBatch tb;
    for (size_t j = 0; j < m; j++) tb.d[j] = tail_desc(B, *slots[j]);
    for (size_t j = m; j < TAIL_BATCH; j++) tb.d[j] = tb.d[0];
    int rc = launch_tails(tb, mm, slots[0]->tasks1, nb, B->logNh, B->logNl, B->seg_log, s);
    if (rc) return rc;
    for (size_t j = 0; j < m; j++) {
      Slot& S = *slots[j];
      H2_HIP(hipEventRecord(S.tail_done, s));
      S.tail_pending = true;
      S.tail_ever = true;
    }
    return H2MI_OK;
  }
  if (g_deferred.size() >= (size_t)std::max(1, B->nslot / 2)) return flush_tails();
  return H2MI_OK;
}

// the small path's batch: one digit launch for the m MSMs; their accumulate / final pair is deferred and batched as ever
static SmallDesc small_desc(const Bases* B, const Slot& S);
static int launch_small(const SmallBatch& sb, uint32_t count, uint32_t max_parts, hipStream_t t);
static int msm_small_batch(Bases* B, const void* const* d_scalars, size_t m, size_t n, void* d_out, hipStream_t s, bool inorder) {
  SmallHeadBatch hb;
  memset(&hb, 0, sizeof(hb));
  Slot* slots[HEAD_BATCH];
  for (size_t j = 0; j < m; j++) {
    Slot& S = B->slot[B->next_slot];
    B->last_slot = B->next_slot;
    B->next_slot = (B->next_slot + 1) % B->nslot;
    if (S.tail_deferred) {
      int rc = flush_tails();
      if (rc) return rc;
    }
    if (S.tail_ever && (S.tail_pending || S.last_stream != s)) H2_HIP(hipStreamWaitEvent(s, S.tail_done, 0));
    if (S.head_pending) H2_HIP(hipStreamWaitEvent(s, S.head_done, 0));
    if (S.accum_pending) H2_HIP(hipStreamWaitEvent(s, S.accum_done, 0));
    S.tail_pending = S.accum_pending = S.head_pending = false;
    S.last_stream = s;
    S.d_out = (char*)d_out + 96 * j;
    S.small_n = (uint32_t)n;
    hb.scalars[j] = (const fe*)d_scalars[j];
    hb.dig[j] = S.sdig;
    slots[j] = &S;
  }
  for (size_t j = m; j < HEAD_BATCH; j++) { hb.scalars[j] = hb.scalars[0]; hb.dig[j] = hb.dig[0]; }
  H2_LAUNCH("k_msm_small_digits", k_msm_small_digits_b, dim3(ceil_div_u32(n, 64), (uint32_t)m), 64, 0, s, hb, (uint32_t)n, (uint32_t)B->n, B->sc, B->sW);
  if (inorder) {  // accumulate + final for the batch at once (on the small path they run on `s` anyway: nothing to defer for)
    SmallBatch sb;
    uint32_t max_parts = 0;
    for (size_t j = 0; j < m; j++) {
      sb.d[j] = small_desc(B, *slots[j]);
      max_parts = std::max(max_parts, sb.d[j].G);
    }
    for (size_t j = m; j < SMALL_BATCH; j++) sb.d[j] = sb.d[0];
    int rc = launch_small(sb, (uint32_t)m, max_parts, s);
    if (rc) return rc;
    for (size_t j = 0; j < m; j++) {
      Slot& S = *slots[j];
      H2_HIP(hipEventRecord(S.tail_done, s));
      S.tail_pending = true;
      S.tail_ever = true;
    }
    return H2MI_OK;
  }
  for (size_t j = 0; j < m; j++) {
    Slot& S = *slots[j];
    H2_HIP(hipEventRecord(S.accum_done, s));  // "inputs consumed": what the deferred pair waits for
    S.accum_pending = true;
    S.accum_ever = true;
    S.tail_deferred = true;
    S.small_deferred = true;
    g_deferred.push_back({B, &S});
  }
  if (g_deferred.size() >= (size_t)std::max(1, B->nslot / 2)) return flush_tails();
  return H2MI_OK;
}

// `inorder`: partition, accumulation and bucket reduction one after the other on `s`, nothing deferred and no stream hops — for a LONE
// commitment whose point the caller reads next (h2mi_msm_bn254_g1_phase_dev with H2MI_MSM_INORDER): the three-stream split buys overlap between consecutive
// MSMs and costs a lone one ~50 us of event hops (2^20: 1645 -> 1580 us, 2^16: 386 -> 336)
static int msm_dev(Bases* B, const void* d_scalars, size_t n, void* d_out, hipStream_t s, bool inorder, bool general) {
  B->last_small = take_small(B, general);
  B->since_join++;
  if (B->last_small) return msm_small(B, d_scalars, n, d_out, s, inorder);
  const uint32_t nb = B->nb, W = B->W;
  // dominant-value shift: only when the MSM covers every registered base (the extra base is their sum)
  const bool shifted = B->has_sum && n == B->n;
  const size_t n_eff = n + (shifted ? 1 : 0);
  const uint32_t total = (uint32_t)(n_eff * W);
  const bool pipelined = !inorder && (s == ctx().stream) && ctx().tail_stream && !getenv("H2MI_MSM_NO_PIPELINE");
  Slot& S = B->slot[B->next_slot];
  B->last_slot = B->next_slot;
  B->next_slot = (B->next_slot + 1) % B->nslot;
  if (S.tail_deferred) {  // every slot of this handle is waiting for its reduction
    int rc = flush_tails();
    if (rc) return rc;
  }
  // streams: caller-provided stream => everything in order on it.  Library stream => three stages:
  //   s : partition level 1: bin count, offsets, scatter (the only readers of the caller's scalars)
  //   hs: partition level 2 (per-bin sort, bucket tables), task offsets
  //   as: accumulation                   (never blocks s: NTTs queued on s meanwhile run beside it)
  //   tail stream: fold ... final, batched over the MSMs since the last join (flush_tails)
  hipStream_t hs = s, as = s;
  // chunk length: chosen by k_msm_bin_sort from the entry count unless H2MI_MSM_S0 fixes it; the partial buffers
  // were sized at registration, so an override must not produce more chunks than they hold
  uint32_t s0_fixed = chunk_override();
  while (s0_fixed && s0_fixed < S0_MAX && ((uint64_t)total + s0_fixed - 1) / s0_fixed + nb > B->max_tasks0) s0_fixed++;
  if (s0_fixed && ((uint64_t)total + s0_fixed - 1) / s0_fixed + nb > B->max_tasks0) return H2MI_ERANGE;
  if (pipelined) {
    hs = ctx().head_stream;
    as = ctx().accum_stream;
    // the first partition level (count, scan, scatter) is the only reader of the caller's scalars and stays
    // on s.  It writes bincnt/binbase/bkeys/vals[0], last read by this slot's previous k_msm_bin_sort.
    if (S.head_pending) H2_HIP(hipStreamWaitEvent(s, S.head_done, 0));
    // the slot's previous MSM ran start to end on a caller's stream: its last kernel recorded tail_done there
    if (S.tail_pending && S.last_stream != s) H2_HIP(hipStreamWaitEvent(s, S.tail_done, 0));
  } else {
    // a caller's stream is ordered against nothing the library did on its own streams (h2mi_join only makes the
    // library stream wait), so it waits for every event this slot has ever recorded; a completed event is free
    if (S.head_ever) H2_HIP(hipStreamWaitEvent(s, S.head_done, 0));
    if (S.accum_ever) H2_HIP(hipStreamWaitEvent(s, S.accum_done, 0));
    if (S.tail_ever) H2_HIP(hipStreamWaitEvent(s, S.tail_done, 0));
  }
  S.last_stream = s;
  const uint32_t ntiles = ceil_div_u32(n_eff, P1_TS);
  const fe* shift = shifted ? S.shift : nullptr;
  static bool attr_set_dev[16] = {};  // function attributes are per device (every shard of a sharded handle launches these)
  bool& attr_set = attr_set_dev[ctx().cur & 15];
  if ((size_t)P1_TS * W * 6 > 150 * 1024) return H2MI_ERANGE;
  if (!attr_set) {  // the scatter kernel stages up to 144 KiB of pairs in LDS (W = 24; 90 KiB at W = 15)
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter<13>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter<15>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter<17>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter<18>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter<19>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_bin_scatter<20>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  {
    // the partition's first level reads the scalars twice (count, scatter): both stay on s, so work queued
    // on s after this call may overwrite them
#define H2_BIN_COUNT(CT) \
  H2_LAUNCH("k_msm_bin_count", k_msm_bin_count<CT>, ntiles, P1_TS, 0, s, (const fe*)d_scalars, n, shift, B->c, W, B->lb, B->nbins, ntiles, S.bincnt, S.tile_live)
#define H2_BIN_SCATTER(CT)                                                                                                                        \
  H2_LAUNCH("k_msm_bin_scatter", k_msm_bin_scatter<CT>, ntiles, P1_TS, (size_t)P1_TS * W * 6, s, (const fe*)d_scalars, n, shift, B->stride, B->c, W, \
            B->lb, B->nbins, ntiles, (const uint32_t*)S.binbase, S.vals[0], S.bkeys, (const uint32_t*)S.tile_live)
    if (shifted) H2_LAUNCH("k_msm_pick_shift", k_msm_pick_shift, 1, 64, 0, s, (const fe*)d_scalars, n, S.shift);
    switch (B->c) {
      case 13: H2_BIN_COUNT(13); break;
      case 15: H2_BIN_COUNT(15); break;
      case 16: H2_BIN_COUNT(16); break;
      case 17: H2_BIN_COUNT(17); break;
      case 18: H2_BIN_COUNT(18); break;
      case 19: H2_BIN_COUNT(19); break;
      case 20: H2_BIN_COUNT(20); break;
      default: H2_BIN_COUNT(0); break;
    }
    const uint32_t cells = B->nbins * ntiles;  // a multiple of 4 (checked at registration)
    const uint32_t nseg = ceil_div_u32(cells, SCAN_SEG_BINS);
    if (nseg > 1) H2_LAUNCH("k_scan_segsum", k_scan_segsum<SCAN_SEG_BINS>, nseg, 1024, 0, s, (const uint32_t*)S.bincnt, cells, S.binseg);
    H2_LAUNCH("k_scan_seg_bins", k_scan_seg<SCAN_SEG_BINS>, dim3(nseg, 1), 1024, 0, s, (const uint32_t*)S.bincnt, S.binbase, (const uint32_t*)nullptr,
              (uint32_t*)nullptr, cells, (const uint32_t*)S.binseg);
    switch (B->c) {
      case 13: H2_BIN_SCATTER(13); break;
      case 15: H2_BIN_SCATTER(15); break;
      case 16: H2_BIN_SCATTER(16); break;
      case 17: H2_BIN_SCATTER(17); break;
      case 18: H2_BIN_SCATTER(18); break;
      case 19: H2_BIN_SCATTER(19); break;
      case 20: H2_BIN_SCATTER(20); break;
      default: H2_BIN_SCATTER(0); break;
    }
#undef H2_BIN_COUNT
#undef H2_BIN_SCATTER
  }
  if (pipelined) {
    H2_HIP(hipEventRecord(S.input_ready, s));
    H2_HIP(hipStreamWaitEvent(hs, S.input_ready, 0));
    // slot reuse: the sorted buffers are read by this slot's previous accumulation, the partial buffers and
    // bucket offsets by its previous tail
    if (S.accum_pending) H2_HIP(hipStreamWaitEvent(hs, S.accum_done, 0));
    if (S.tail_pending) H2_HIP(hipStreamWaitEvent(hs, S.tail_done, 0));
  }
  S.tail_pending = false;
  S.accum_pending = false;
  S.head_pending = false;
  if (B->seg_log) {
    H2_LAUNCH("k_msm_bin_sort", k_msm_bin_sort_wide, B->nbins, P2_THREADS, 0, hs, (const uint16_t*)S.bkeys, (const uint32_t*)S.vals[0],
              (const uint32_t*)S.binbase, ntiles, B->nbins, B->lb, s0_fixed, nb, S.vals[1], S.off, S.np[0], S.np[1], S.s0_dev);
    const uint32_t tsegs = ceil_div_u32(nb, SCAN_SEG_BINS);  // nb > 65536 task counters: segmented, one array per launch
    for (int i = 0; i < 2; i++) {
      H2_LAUNCH("k_scan_segsum", k_scan_segsum<SCAN_SEG_BINS>, tsegs, 1024, 0, hs, (const uint32_t*)S.np[i], nb, S.tseg[i]);
      H2_LAUNCH("k_scan_seg_tasks", k_scan_seg<SCAN_SEG_BINS>, dim3(tsegs, 1), 1024, 0, hs, (const uint32_t*)S.np[i], S.toff[i], (const uint32_t*)nullptr,
                (uint32_t*)nullptr, nb, (const uint32_t*)S.tseg[i]);
    }
  } else {
    H2_LAUNCH("k_msm_bin_sort", k_msm_bin_sort, B->nbins, P2_THREADS, 0, hs, (const uint8_t*)S.bkeys, (const uint32_t*)S.vals[0],
              (const uint32_t*)S.binbase, ntiles, B->nbins, B->lb, s0_fixed, nb, S.vals[1], S.off, S.np[0], S.np[1], S.s0_dev);
    H2_LAUNCH("k_scan_seg_tasks", k_scan_seg<SCAN_SEG_TASKS>, dim3(1, 2), 1024, 0, hs, (const uint32_t*)S.np[0], S.toff[0], (const uint32_t*)S.np[1], S.toff[1], nb,
              (const uint32_t*)nullptr);
  }
  // upper bound of the chunks (zero digits leave no entry): whole rounds of the resident grid, or total / override
  // (a chunk holds at least one entry: small base sets never come near a whole round)
  const uint32_t chunks0 = s0_fixed ? (uint32_t)(((uint64_t)total + s0_fixed - 1) / s0_fixed) : std::min(accum_rounds(total) * ACCUM_RESIDENT_CHUNKS, total);
  const uint32_t tasks0 = chunks0 + nb;                                             // upper bound of the partial sums
  if (pipelined) {
    H2_HIP(hipEventRecord(S.head_done, hs));
    H2_HIP(hipStreamWaitEvent(as, S.head_done, 0));
    S.head_pending = true;
    S.head_ever = true;
  }
  // Occupancy cap: the accumulation runs as fast with 2 wavefronts per SIMD as with 4 (it is bound by VALU
  // issue, not latency), but at 4 it owns every VGPR of the chip and the short kernels of the neighbouring
  // MSMs (bucket reduction, partition) cannot start until it drains.  An unused dynamic-LDS reservation of
  // 56000 B holds it at two workgroups per CU and leaves half the registers and 48 KB of LDS per CU free:
  // back-to-back MSMs 2^20: 1.90 -> 1.74 ms.  H2MI_ACCUM_LDS=0 removes the cap.
  static const size_t accum_lds = ab_env("H2MI_ACCUM_LDS") ? (size_t)atoi(ab_env("H2MI_ACCUM_LDS")) : 56000;
#ifdef H2MI_AB
  static bool prio_set = false;
  if (!prio_set && ab_env("H2MI_AB_PRIO")) {
    const uint32_t one = 1;
    H2_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ab_prio), &one, 4));
  }
  prio_set = true;
  // pair-affine accumulation (experiment, -DH2MI_AB library only): from H2MI_MSM_PA_MIN entries (default 2^22)
  static const bool pa_on = ab_env("H2MI_MSM_PA") != nullptr;
  static const uint32_t pa_min = ab_env("H2MI_MSM_PA_MIN") ? (uint32_t)atoll(ab_env("H2MI_MSM_PA_MIN")) : (1u << 22);
  if (pa_on && !s0_fixed && total >= pa_min) {
    if (S.pa_T < chunks0) {
      if (S.pa_spill) { H2_HIP(hipFree(S.pa_spill)); H2_HIP(hipFree(S.pa_tot)); S.pa_spill = S.pa_tot = nullptr; S.pa_T = 0; }
      const uint32_t cap = std::min(accum_rounds((uint32_t)(B->stride * W)) * ACCUM_RESIDENT_CHUNKS, (uint32_t)(B->stride * W));
      H2_HIP(hipMalloc((void**)&S.pa_spill, (size_t)PA_MAX_PAIRS * 9 * 4 * cap));
      H2_HIP(hipMalloc((void**)&S.pa_tot, (size_t)9 * 4 * cap));
      S.pa_T = cap;
    }
    // the forward pass and the inversions belong to the head (they run beside the previous MSM's accumulation)
    H2_LAUNCH("k_msm_pa_forward", k_msm_pa_forward, ceil_div_u32(chunks0, 256), 256, 0, hs, (const uint32_t*)S.vals[1], (const uint32_t*)S.off, nb,
              (const uint32_t*)S.s0_dev, (const uint8_t*)B->table, chunks0, S.pa_spill, S.pa_tot);
    H2_LAUNCH("k_msm_pa_invert", k_msm_pa_invert, ceil_div_u32(chunks0, 1024), 256, 0, hs, S.pa_tot, chunks0);
    if (pipelined) {
      H2_HIP(hipEventRecord(S.head_done, hs));
      H2_HIP(hipStreamWaitEvent(as, S.head_done, 0));
    }
    H2_LAUNCH("k_msm_accum", k_msm_pa_backward, ceil_div_u32(chunks0, 256), 256, accum_lds, as, (const uint32_t*)S.vals[1], (const uint32_t*)S.off,
              (const uint32_t*)S.toff[0], nb, (const uint32_t*)S.s0_dev, (const uint8_t*)B->table, S.part[0], chunks0, (const uint32_t*)S.pa_spill,
              (const uint32_t*)S.pa_tot);
  } else
#endif
  H2_LAUNCH("k_msm_accum", k_msm_accum, ceil_div_u32(chunks0, 256), 256, accum_lds, as, (const uint32_t*)S.vals[1], (const uint32_t*)S.off,
            (const uint32_t*)S.toff[0], nb, (const uint32_t*)S.s0_dev, (const uint8_t*)B->table, S.part[0]);
  S.d_out = d_out;
  S.tasks1 = tasks0 / S1 + nb;
  if (pipelined) {
    // the bucket reduction is deferred: flush_tails() runs it for every MSM queued since the last join
    H2_HIP(hipEventRecord(S.accum_done, as));
    S.accum_pending = true;
    S.accum_ever = true;
    S.tail_deferred = true;
    g_deferred.push_back({B, &S});
    static const bool eager = ab_env("H2MI_MSM_EAGER_TAIL") != nullptr;  // A/B: one reduction per MSM, at once
    // half the slots: the reductions of one half run beside the accumulation of the other
    if (eager || g_deferred.size() >= (size_t)std::max(1, B->nslot / 2)) return flush_tails();
    return H2MI_OK;
  }
  TailBatch tb;
  for (uint32_t j = 0; j < TAIL_BATCH; j++) tb.d[j] = tail_desc(B, S);
  int rc = launch_tails(tb, 1, S.tasks1, nb, B->logNh, B->logNl, B->seg_log, s);
  if (rc) return rc;
  H2_HIP(hipEventRecord(S.tail_done, s));
  S.tail_pending = true;
  S.tail_ever = true;
  return H2MI_OK;
}

// ---- small base sets: host side ---------------------------------------------------------------------------------
static SmallDesc small_desc(const Bases* B, const Slot& S) {
  SmallDesc d;
  d.dig = S.sdig; d.table = B->stable; d.part = S.spart; d.cnt = S.scnt; d.ticket_at = B->sG; d.out = (uint8_t*)S.d_out; d.stats = S.stats;
  d.n = S.small_n; d.n_reg = (uint32_t)B->n; d.W = B->sW; d.c = B->sc; d.r = B->sr; d.lanes = B->slanes; d.canonical = g_canonical ? 1u : 0u;
  d.G = ceil_div_u32((uint64_t)d.n * d.W, (uint64_t)d.lanes * d.r);  // <= B->sG: an MSM over the first n <= n_reg bases
  return d;
}

static int launch_small(const SmallBatch& sb, uint32_t count, uint32_t max_parts, hipStream_t t) {
  constexpr size_t LDS = (size_t)SMALL_PARTS_MAX * PART_BYTES;  // 72 KB: above the 64 KB a kernel gets without asking
  static bool attr_set_dev[16] = {};  // function attributes are per device (every shard of a sharded handle launches these)
  bool& attr_set = attr_set_dev[ctx().cur & 15];
  if (!attr_set) {
    H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_small_accum), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    attr_set = true;
  }
  H2_LAUNCH("k_msm_small_accum", k_msm_small_accum, dim3(max_parts, count), SMALL_THREADS, LDS, t, sb);
  return H2MI_OK;
}

// One MSM over the first n bases of a small base set.  The digit kernel runs on the caller's stream (it is the only reader of
// the scalars: work queued on s afterwards may overwrite them); on the library stream the accumulate + final pair is deferred
// to the next join / flush and batched with the other small MSMs of the phase, on a caller's stream it follows in order.
static int msm_small(Bases* B, const void* d_scalars, size_t n, void* d_out, hipStream_t s, bool inorder) {
  const bool pipelined = !inorder && (s == ctx().stream) && ctx().tail_stream && !getenv("H2MI_MSM_NO_PIPELINE");
  Slot& S = B->slot[B->next_slot];
  B->last_slot = B->next_slot;
  B->next_slot = (B->next_slot + 1) % B->nslot;
  if (S.tail_deferred) {  // every slot of this handle is waiting for its reduction
    int rc = flush_tails();
    if (rc) return rc;
  }
  // slot reuse: the digit bytes and partial sums are read by the slot's previous accumulate / final pair — and a slot last
  // used by the general pipeline (H2MI_MSM_GENERAL, or the streaming rule of take_small) still owes its head / accumulation events
  if (S.tail_ever && (S.tail_pending || !pipelined || S.last_stream != s)) H2_HIP(hipStreamWaitEvent(s, S.tail_done, 0));
  if (S.head_pending) H2_HIP(hipStreamWaitEvent(s, S.head_done, 0));
  if (S.accum_pending) H2_HIP(hipStreamWaitEvent(s, S.accum_done, 0));
  S.tail_pending = S.accum_pending = S.head_pending = false;
  S.last_stream = s;
  S.d_out = d_out;
  S.small_n = (uint32_t)n;
  H2_LAUNCH("k_msm_small_digits", k_msm_small_digits, ceil_div_u32(n, 64), 64, 0, s, (const fe*)d_scalars, (uint32_t)n, (uint32_t)B->n, B->sc, B->sW, S.sdig);
  if (pipelined) {
    H2_HIP(hipEventRecord(S.accum_done, s));  // "inputs consumed": what the deferred pair waits for
    S.accum_pending = true;
    S.accum_ever = true;
    S.tail_deferred = true;
    S.small_deferred = true;
    g_deferred.push_back({B, &S});
    static const bool eager = ab_env("H2MI_MSM_EAGER_TAIL") != nullptr;
    if (eager || g_deferred.size() >= (size_t)std::max(1, B->nslot / 2)) return flush_tails();
    return H2MI_OK;
  }
  SmallBatch sb;
  for (uint32_t j = 0; j < SMALL_BATCH; j++) sb.d[j] = small_desc(B, S);
  int rc = launch_small(sb, 1, sb.d[0].G, s);
  if (rc) return rc;
  H2_HIP(hipEventRecord(S.tail_done, s));
  S.tail_pending = true;
  S.tail_ever = true;
  return H2MI_OK;
}

// one fold level (<= S1 partials per task: part[0] -> part[1]), then FG lanes per bucket finish into the
// dense array.  A bucket holding m points leaves ceil(m / (s0 * S1)) partials for the finish kernel: 1 in
// the uniform case at k = 20, <= ~50 for the hot 0/1 buckets of witness-like columns (7 serial additions
// per lane), n / (s0 * S1) if every scalar is the same (slow but correct).  Then the weighted bucket sum.
static int launch_tails(const TailBatch& tb, uint32_t count, uint32_t max_tasks1, uint32_t max_nb, uint32_t max_logNh, uint32_t max_logNl,
                        uint32_t max_seg, hipStream_t t) {
  // fold / finish: quads (4 lanes per point operation) while the grid stays latency-bound, single lanes beyond
  const bool force_lane = ab_env("H2MI_MSM_TAIL_LANES") != nullptr;  // A/B
  bool none_folds = true;
  for (uint32_t j = 0; j < count; j++) none_folds = none_folds && tb.d[j].no_fold != 0;
  if (none_folds) {
    // no fold level at all
  } else if (!force_lane && (uint64_t)max_tasks1 * count <= 65536) {
    H2_LAUNCH("k_msm_fold", k_msm_fold<true>, dim3(ceil_div_u32((uint64_t)max_tasks1 * 4, 256), count), 256, 0, t, tb);
  } else {
    H2_LAUNCH("k_msm_fold", k_msm_fold<false>, dim3(ceil_div_u32(max_tasks1, 256), count), 256, 0, t, tb);
  }
  const uint32_t hot_blocks = ceil_div_u32(max_nb, 256);  // hot-bucket finishers: the first workgroups of the finish launch
  const size_t hot_lds = 64 * PART_BYTES;
  if (max_seg) {
    H2_LAUNCH("k_msm_finish", (k_msm_finish<false, FG_WIDE>), dim3(hot_blocks + ceil_div_u32((uint64_t)max_nb * FG_WIDE, 256), count), 256, hot_lds, t, tb, hot_blocks);
  } else if (!force_lane && (uint64_t)max_nb * FG_NARROW * count <= 65536) {
    H2_LAUNCH("k_msm_finish", (k_msm_finish<true, FG_NARROW>), dim3(hot_blocks + ceil_div_u32((uint64_t)max_nb * FG_NARROW * 4, 256), count), 256, hot_lds, t, tb,
              hot_blocks);
  } else {
    H2_LAUNCH("k_msm_finish", (k_msm_finish<false, FG_NARROW>), dim3(hot_blocks + ceil_div_u32((uint64_t)max_nb * FG_NARROW, 256), count), 256, hot_lds, t, tb,
              hot_blocks);
  }
  if (max_seg) H2_LAUNCH("k_msm_seg", k_msm_seg, dim3((4u << MAT_LOG) / 256, count), 256, 0, t, tb);
  H2_LAUNCH("k_msm_rowcol", k_msm_rowcol, dim3(((max_seg ? 2u : 1u) << max_logNh) + (1u << max_logNl), count), TAIL_THREADS, 256 * PART_BYTES, t, tb);
  H2_LAUNCH("k_msm_weighted", k_msm_weighted, dim3(max_logNh + max_logNl + 1 + (max_seg ? 1 : 0), count), TAIL_THREADS, 256 * PART_BYTES, t, tb);
  H2_LAUNCH("k_msm_final", k_msm_final, count, 128, 32 * PART_BYTES, t, tb);
  return H2MI_OK;
}

static TailDesc tail_desc(const Bases* B, const Slot& S) {
  TailDesc d;
  d.part0 = S.part[0];
  d.toff0 = S.toff[0]; d.np0 = S.np[0]; d.toff1 = S.toff[1]; d.np1 = S.np[1]; d.off = S.off;
  d.part1 = S.part[1]; d.dense = S.dense; d.rc = S.rc; d.g = S.g; d.out = (uint8_t*)S.d_out;
  d.stats = S.stats;
  d.nb = B->nb; d.logNh = B->logNh; d.logNl = B->logNl; d.canonical = g_canonical ? 1u : 0u;
  d.seg_log = B->seg_log; d.dense2 = S.dense2; d.vsum = S.vsum; d.mat = B->seg_log ? S.dense2 : S.dense;
  d.hot_min = B->seg_log ? HOT_MIN_WIDE : HOT_MIN;
  // wide windows: 2^17 .. 2^19 buckets hold a partial or two each, the fold level would copy them (0.08 - 0.18 ms per MSM, measured);
  // the finish kernels read the accumulation's partials directly.  (Round 4 tried the same for up to 2^14 / 2^15 buckets, where the
  // chain is what a join waits for: a lone 2^16 MSM gains ~10 us, a phase of four loses 4 - 7 %: profiles/r04_msm_mid_sweep.txt.)
  d.no_fold = B->seg_log ? 1u : 0u;
  if (d.no_fold) { d.part1 = S.part[0]; d.toff1 = S.toff[0]; d.np1 = S.np[0]; }
  return d;
}

// launch the deferred bucket reductions (all handles) as batches on the tail stream
static int flush_tails() {
  if (g_deferred.empty()) return H2MI_OK;
  const int prev = ctx().cur;
  // the reductions of a device run on that device's tail stream, batched per device
  for (int dev = 0; dev < (int)ctx().devs.size(); dev++) {
    std::vector<Deferred> mine;
    for (const Deferred& d : g_deferred)
      if (d.B->dev == dev) mine.push_back(d);
    if (mine.empty()) continue;
    int rc0 = use_device(dev);
    if (rc0) return rc0;
    hipStream_t t = ctx().tail_stream;
    // three kinds of deferred work, each batched with its own kind: the small path's accumulate + final pairs, and the bucket
    // reductions of narrow and of wide windows (a wide descriptor in a batch used to switch every member to the two-lane
    // finish: correct, but a silent latency cliff for mixed-size commitment streams — round-3 ADVICE)
    std::vector<Deferred> group[3];
    for (const Deferred& d : mine) group[d.S->small_deferred ? 0 : d.B->seg_log ? 2 : 1].push_back(d);
    for (size_t i = 0; i < group[0].size();) {
      SmallBatch sb;
      uint32_t count = 0, max_parts = 0;
      const size_t first = i;
      // the small path's pair runs on the device's LIBRARY stream, behind the digit kernels that were queued there: a
      // cross-stream hop costs ~10 us each way (event record -> wait -> launch, measured: 26 us of gaps around 81 us of
      // kernels), more than these kernels could ever gain from running beside the stream's other work
      hipStream_t ls = ctx().stream;
      for (; i < group[0].size() && count < SMALL_BATCH; i++, count++) {
        Bases* B = group[0][i].B;
        Slot& S = *group[0][i].S;
        sb.d[count] = small_desc(B, S);
        max_parts = std::max(max_parts, sb.d[count].G);
      }
      for (uint32_t j = count; j < SMALL_BATCH; j++) sb.d[j] = sb.d[0];  // never read: blockIdx.y < count
      int rc = launch_small(sb, count, max_parts, ls);
      if (rc) return rc;
      for (size_t j = first; j < i; j++) {
        Slot& S = *group[0][j].S;
        H2_HIP(hipEventRecord(S.tail_done, ls));
        S.tail_pending = true;
        S.tail_ever = true;
        S.tail_deferred = false;
        S.small_deferred = false;
      }
    }
    for (int gi = 1; gi < 3; gi++) {
      const std::vector<Deferred>& grp = group[gi];
      size_t i = 0;
      while (i < grp.size()) {
        TailBatch tb;
        uint32_t count = 0, max_tasks1 = 0, max_nb = 0, max_logNh = 0, max_logNl = 0, max_seg = 0;
        const size_t first = i;
        for (; i < grp.size() && count < TAIL_BATCH; i++, count++) {
          Bases* B = grp[i].B;
          Slot& S = *grp[i].S;
          H2_HIP(hipStreamWaitEvent(t, S.accum_done, 0));
          tb.d[count] = tail_desc(B, S);
          max_tasks1 = std::max(max_tasks1, S.tasks1);
          max_nb = std::max(max_nb, B->nb);
          max_logNh = std::max(max_logNh, B->logNh);
          max_logNl = std::max(max_logNl, B->logNl);
          max_seg = std::max(max_seg, B->seg_log);
        }
        for (uint32_t j = count; j < TAIL_BATCH; j++) tb.d[j] = tb.d[0];  // never read: blockIdx.y < count
        int rc = launch_tails(tb, count, max_tasks1, max_nb, max_logNh, max_logNl, max_seg, t);
        if (rc) return rc;
        for (size_t j = first; j < i; j++) {
          Slot& S = *grp[j].S;
          H2_HIP(hipEventRecord(S.tail_done, t));
          S.tail_pending = true;
          S.tail_ever = true;
          S.tail_deferred = false;
        }
      }
    }
  }
  g_deferred.clear();
  return use_device(prev);
}

static const uint64_t G1_IDENTITY_J[12] = {0, 0, 0, 0, 0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL, 0, 0, 0, 0};

static hipError_t copy_between(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t s) {
  const int pd = ctx().devs[(size_t)dst_dev].device, ps = ctx().devs[(size_t)src_dev].device;
  if (pd == ps) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
  return hipMemcpyPeerAsync(dst, pd, src, ps, bytes, s);
}

// gather the partial results of every sharded MSM queued since the last join to the primary device and fold them;
// the primary stream s0 has already been made to wait for every reduction (msm_join_all)
static int fold_sharded(hipStream_t s0) {
  if (g_pending_folds.empty()) return H2MI_OK;
  for (const PendingFold& pf : g_pending_folds) {
    Sharded* sh = pf.sh;
    const size_t W = sh->shard.size();
    uint8_t* g = sh->gather + (size_t)pf.ring * W * 96;
    for (size_t i = 0; i < W; i++) {
      const Shard& sd = sh->shard[i];
      const int dev = g_bases[sd.handle]->dev;
      H2_HIP(copy_between(g + 96 * i, 0, sd.part + (size_t)pf.ring * 96, dev, 96, s0));
    }
    int rc = launch_fold_groups(g, W, 1, (uint8_t*)pf.d_out, s0);
    if (rc) return rc;
  }
  g_pending_folds.clear();
  return H2MI_OK;
}

static void free_sharded(Sharded* sh) {
  for (Shard& sd : sh->shard) {
    H2_IGNORE(hipFree(sd.stage));
    H2_IGNORE(hipFree(sd.part));
    if (sd.ready) H2_IGNORE(hipEventDestroy(sd.ready));
  }
  H2_IGNORE(hipFree(sh->gather));
  delete sh;
}

// register bases (host pointer, or device pointer on the primary device) as one slice per device
static int register_sharded(const void* bases, bool on_host, size_t n, uint64_t* handle_out) {
  const size_t D = ctx().devs.size();
  Sharded* sh = new Sharded();
  sh->n = n;
  int rc = H2MI_OK;
  for (size_t i = 0; i < D && !rc; i++) {
    const size_t base = n / D, rem = n % D;
    const size_t lo = i * base + std::min(i, rem), hi = lo + base + (i < rem ? 1 : 0);
    if (hi == lo) continue;  // fewer bases than devices
    rc = use_device((int)i);
    if (rc) break;
    hipStream_t s = ctx().stream;
    Shard sd;
    sd.lo = lo;
    sd.hi = hi;
    void* tmp = nullptr;
    if (hipMalloc(&tmp, (hi - lo) * 64) != hipSuccess || hipMalloc((void**)&sd.part, (size_t)SHARD_RING * 96) != hipSuccess ||
        hipEventCreateWithFlags(&sd.ready, hipEventDisableTiming) != hipSuccess) {
      H2_IGNORE(hipFree(tmp));
      H2_IGNORE(hipFree(sd.part));
      rc = H2MI_ENOMEM;
      break;
    }
    hipError_t e = on_host ? hipMemcpyAsync(tmp, (const uint8_t*)bases + lo * 64, (hi - lo) * 64, hipMemcpyHostToDevice, s)
                           : copy_between(tmp, (int)i, (const uint8_t*)bases + lo * 64, 0, (hi - lo) * 64, s);
    if (e != hipSuccess) rc = H2MI_EHIP;
    if (!rc) rc = register_dev(tmp, hi - lo, &sd.handle, s);  // synchronises s
    H2_IGNORE(hipFree(tmp));
    if (rc) {
      H2_IGNORE(hipFree(sd.part));
      H2_IGNORE(hipEventDestroy(sd.ready));
      break;
    }
    sh->shard.push_back(sd);
  }
  if (!rc) rc = use_device(0);
  if (!rc && hipMalloc((void**)&sh->gather, (size_t)SHARD_RING * sh->shard.size() * 96) != hipSuccess) rc = H2MI_ENOMEM;
  if (rc) {
    for (Shard& sd : sh->shard) h2mi_bases_release(sd.handle);
    use_device(0);
    free_sharded(sh);
    return rc;
  }
  const uint64_t h = g_next_handle++;
  g_sharded[h] = sh;
  *handle_out = h;
  return H2MI_OK;
}

// queue one MSM over the first n registered bases of a sharded handle; the folded result reaches d_out (primary
// device) at the next join.  scalars: host pointer, or device pointer on the primary device.
static int msm_sharded(Sharded* sh, const void* scalars, bool on_host, size_t n, void* d_out) {
  if (n > sh->n) return H2MI_ERANGE;
  size_t queued = 0;
  for (const PendingFold& pf : g_pending_folds) queued += pf.sh == sh ? 1 : 0;
  int rc = use_device(0);
  if (rc) return rc;
  hipStream_t s0 = ctx().stream;
  if (queued >= SHARD_RING) {  // the ring of partial results is full: fold what is queued
    rc = msm_join_all(s0);
    if (rc) return rc;
  }
  const uint32_t ring = sh->next++ % SHARD_RING;
  hipEvent_t src_ready = nullptr;
  if (!on_host) {  // the other devices' copies must follow whatever produced the scalars on the primary stream
    src_ready = sh->shard[0].ready;
    H2_HIP(hipEventRecord(src_ready, s0));
  }
  for (size_t i = 0; i < sh->shard.size() && !rc; i++) {
    Shard& sd = sh->shard[i];
    Bases* B = g_bases[sd.handle];
    rc = use_device(B->dev);
    if (rc) break;
    hipStream_t si = ctx().stream;
    const size_t cnt = n > sd.lo ? std::min(n, sd.hi) - sd.lo : 0;
    uint8_t* out = sd.part + (size_t)ring * 96;
    if (cnt == 0) {
      // nothing of this MSM falls into the shard: its partial result is the identity.  msm_join_all only waits for the
      // reductions of slots that ran an MSM, so the primary stream is ordered behind this write explicitly
      H2_HIP(hipMemcpyAsync(out, G1_IDENTITY_J, 96, hipMemcpyHostToDevice, si));
      H2_HIP(hipEventRecord(sd.ready, si));
      H2_HIP(hipStreamWaitEvent(s0, sd.ready, 0));
      continue;
    }
    const uint8_t* src = (const uint8_t*)scalars + sd.lo * 32;
    const void* d_sc = src;
    if (on_host || B->dev != 0) {
      if (!sd.stage && hipMalloc((void**)&sd.stage, (sd.hi - sd.lo) * 32) != hipSuccess) {
        sd.stage = nullptr;
        rc = H2MI_ENOMEM;
        break;
      }
      if (on_host) {
        H2_HIP(hipMemcpyAsync(sd.stage, src, cnt * 32, hipMemcpyHostToDevice, si));
      } else {
        H2_HIP(hipStreamWaitEvent(si, src_ready, 0));
        H2_HIP(copy_between(sd.stage, B->dev, src, 0, cnt * 32, si));
        // the caller may overwrite its scalars in primary-stream order once this call returns
        if (i != 0) {
          H2_HIP(hipEventRecord(sd.ready, si));
          H2_HIP(hipStreamWaitEvent(s0, sd.ready, 0));
        }
      }
      d_sc = sd.stage;
    }
    rc = msm_dev(B, d_sc, cnt, out, si);
  }
  int rc2 = use_device(0);
  if (rc) return rc;
  if (rc2) return rc2;
  g_pending_folds.push_back({sh, ring, d_out});
  return H2MI_OK;
}

// h2mi_shutdown: every registration (plain, sharded, ad hoc) belongs to the devices that are being torn down
void msm_teardown() {
  g_deferred.clear();
  g_pending_folds.clear();
  for (auto& kv : g_sharded) free_sharded(kv.second);
  g_sharded.clear();
  for (auto& kv : g_bases) {
    if (use_device(kv.second->dev) == H2MI_OK) free_bases(kv.second);
  }
  g_bases.clear();
  g_adhoc.clear();
}

// make stream `s` wait for every outstanding MSM (device-side join, no host synchronisation)
int msm_flush_all() { return flush_tails(); }

int msm_join_all(hipStream_t s) {
  int rc = flush_tails();
  if (rc) return rc;
  for (auto& kv : g_bases) kv.second->since_join = 0;  // the caller has caught up with its queue
  for (auto& kv : g_bases)
    for (Slot& S : kv.second->slot)
      if (S.tail_pending) {
        H2_HIP(hipStreamWaitEvent(s, S.tail_done, 0));
        // only the owning device's library stream is thereby ordered behind this slot's previous use
        if (s == ctx().devs[(size_t)kv.second->dev].stream) S.tail_pending = false;
      }
  if (s == ctx().devs[0].stream) return fold_sharded(s);  // partial results of sharded MSMs: gather + fold on the primary device
  return H2MI_OK;
}

}  // namespace h2

using namespace h2;

extern "C" {

int h2mi_bases_register_dev(const void* d_bases, size_t n, uint64_t* handle_out) {
  H2_REQUIRE_INIT();
  if (!d_bases || !handle_out) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (ctx().devs.size() > 1) return register_sharded(d_bases, /*on_host=*/false, n, handle_out);
  return register_dev(d_bases, n, handle_out, ctx().stream);
}

int h2mi_bases_register(const uint64_t* bases, size_t n, uint64_t* handle_out) {
  H2_REQUIRE_INIT();
  if (!bases || !handle_out || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (ctx().devs.size() > 1) return register_sharded(bases, /*on_host=*/true, n, handle_out);
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, n * 64);
  if (e == hipErrorOutOfMemory) return H2MI_ENOMEM;
  H2_HIP(e);
  int rc = H2MI_OK;
  if (hipMemcpyAsync(d, bases, n * 64, hipMemcpyHostToDevice, ctx().stream) != hipSuccess) rc = H2MI_EHIP;
  if (!rc) rc = register_dev(d, n, handle_out, ctx().stream);
  H2_IGNORE(hipStreamSynchronize(ctx().stream));
  H2_IGNORE(hipFree(d));
  return rc;
}

int h2mi_bases_release(uint64_t handle) {
  H2_REQUIRE_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  auto sh = g_sharded.find(handle);
  if (sh != g_sharded.end()) {
    int rc = msm_join_all(ctx().devs[0].stream);  // folds still queued on it
    if (rc) return rc;
    for (DevCtx& d : ctx().devs) {
      set_thread_device(d.device);
      (void)hipDeviceSynchronize();
    }
    use_device(0);
    for (Shard& sd : sh->second->shard) h2mi_bases_release(sd.handle);
    free_sharded(sh->second);
    g_sharded.erase(sh);
    return H2MI_OK;
  }
  auto it = g_bases.find(handle);
  if (it == g_bases.end()) return H2MI_EHANDLE;
  flush_tails();
  if (use_device(it->second->dev) == H2MI_OK) (void)hipDeviceSynchronize();  // the device that owns the tables
  free_bases(it->second);
  use_device(0);
  g_bases.erase(it);
  for (size_t i = 0; i < g_adhoc.size(); i++)
    if (g_adhoc[i].handle == handle) {
      g_adhoc.erase(g_adhoc.begin() + i);
      break;
    }
  return H2MI_OK;
}

int h2mi_bases_info(uint64_t handle, uint32_t* c, uint32_t* windows, uint32_t* buckets, uint64_t* n) {
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  auto sh = g_sharded.find(handle);
  if (sh != g_sharded.end()) {  // window geometry of the first slice, base count of the whole set
    int rc = h2mi_bases_info(sh->second->shard[0].handle, c, windows, buckets, nullptr);
    if (n) *n = sh->second->n;
    return rc;
  }
  auto it = g_bases.find(handle);
  if (it == g_bases.end()) return H2MI_EHANDLE;
  if (c) *c = it->second->c;
  if (windows) *windows = it->second->W;
  if (buckets) *buckets = it->second->nb;
  if (n) *n = it->second->n;
  return H2MI_OK;
}

// ---- ad-hoc bases (handle = 0): best_multiexp(coeffs, bases) called with a plain slice --------------------------
// Registering builds the W-window table (c doublings and an inversion per point per window: ~45 ms at 2^20, 25 x
// the MSM itself), so an unregistered call must not pay it every time: the bases are uploaded (the price of the
// host-pointer form, 64 B x n over PCIe), fingerprinted ON THE DEVICE over every byte (two independent 64-bit
// multiply-xor sums) and looked up in a small cache of ad-hoc registrations keyed by (n, fingerprint).  A prover that
// keeps calling with the same SRS slice rebuilds nothing; changed bases change the fingerprint and are re-registered.
__global__ void __launch_bounds__(256) k_bases_fingerprint(const uint64_t* words, size_t count, uint64_t* out /* [2], zeroed */) {
  uint64_t a = 0, b = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t w = words[i];
    uint64_t x = (w ^ (i * 0x9E3779B97F4A7C15ull)) * 0xBF58476D1CE4E5B9ull;
    x ^= x >> 29;
    a += x * 0x94D049BB133111EBull;
    uint64_t y = (w + (i + 1) * 0xD6E8FEB86659FD93ull) * 0xFF51AFD7ED558CCDull;
    y ^= y >> 32;
    b ^= y * 0xC4CEB9FE1A85EC53ull + (y >> 17);
  }
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b ^= __shfl_down(b, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd((unsigned long long*)&out[0], (unsigned long long)a);
    atomicXor((unsigned long long*)&out[1], (unsigned long long)b);
  }
}

static int adhoc_handle(const uint64_t* bases, size_t n, uint64_t* handle_out) {
  hipStream_t s = ctx().stream;
  DevMem d, fp;
  hipError_t e = d.alloc(n * 64);
  if (e == hipErrorOutOfMemory) return H2MI_ENOMEM;
  H2_HIP(e);
  H2_HIP(fp.alloc(16));
  H2_HIP(hipMemcpyAsync(d.p, bases, n * 64, hipMemcpyHostToDevice, s));
  H2_HIP(hipMemsetAsync(fp.p, 0, 16, s));
  const uint32_t grid = (uint32_t)std::min<size_t>(1024, (n * 8 + 255) / 256);
  H2_LAUNCH("k_bases_fingerprint", k_bases_fingerprint, grid, 256, 0, s, d.as<uint64_t>(), n * 8, fp.as<uint64_t>());
  uint64_t h[2] = {0, 0};
  H2_HIP(hipMemcpyAsync(h, fp.p, 16, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  g_adhoc_clock++;
  for (AdHoc& a : g_adhoc)
    if (a.n == n && a.fp[0] == h[0] && a.fp[1] == h[1] && g_bases.count(a.handle)) {
      a.last_use = g_adhoc_clock;
      *handle_out = a.handle;
      return H2MI_OK;
    }
  if (g_adhoc.size() >= ADHOC_MAX) {  // drop the least recently used ad-hoc registration
    size_t victim = 0;
    for (size_t i = 1; i < g_adhoc.size(); i++)
      if (g_adhoc[i].last_use < g_adhoc[victim].last_use) victim = i;
    h2mi_bases_release(g_adhoc[victim].handle);
    g_adhoc.erase(g_adhoc.begin() + victim);
  }
  uint64_t nh = 0;
  int rc = register_dev(d.p, n, &nh, s, /*allow_small=*/false);
  if (rc) return rc;
  g_adhoc_builds++;
  g_adhoc.push_back({n, {h[0], h[1]}, nh, g_adhoc_clock});
  *handle_out = nh;
  return H2MI_OK;
}

// G1::identity() = (0, 1, 0) in Montgomery form: what best_multiexp returns for empty slices
static const uint64_t G1_IDENTITY[12] = {0, 0, 0, 0, 0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL, 0, 0, 0, 0};

static int msm_dev_entry(uint64_t handle, const void* d_scalars, size_t n, void* d_out_jacobian, h2mi_stream_t stream, bool inorder, bool general = false);
int h2mi_msm_bn254_g1_dev(uint64_t handle, const void* d_scalars, size_t n, void* d_out_jacobian, h2mi_stream_t stream) {
  return msm_dev_entry(handle, d_scalars, n, d_out_jacobian, stream, false);
}
static int msm_dev_entry(uint64_t handle, const void* d_scalars, size_t n, void* d_out_jacobian, h2mi_stream_t stream, bool inorder, bool general) {
  H2_REQUIRE_INIT();
  if (!d_out_jacobian || (!d_scalars && n != 0)) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  if (n == 0) {  // empty sum
    H2_HIP(hipMemcpyAsync(d_out_jacobian, G1_IDENTITY, 96, hipMemcpyHostToDevice, pick_stream(stream)));
    return H2MI_OK;
  }
  auto sh = g_sharded.find(handle);
  if (sh != g_sharded.end()) {  // every slice from this thread; folded into d_out_jacobian at the next join
    if (stream) return H2MI_EINVAL;  // sharded MSMs run on the library's streams only
    return msm_sharded(sh->second, d_scalars, /*on_host=*/false, n, d_out_jacobian);
  }
  auto it = g_bases.find(handle);
  if (it == g_bases.end()) return H2MI_EHANDLE;
  if (n > it->second->n) return H2MI_ERANGE;
  return msm_dev(it->second, d_scalars, n, d_out_jacobian, pick_stream(stream), inorder, general);
}

static int msm_batch_entry(uint64_t handle, const void* const* d_scalars, size_t count, size_t n, void* d_out_jacobian, h2mi_stream_t stream, bool sparse,
                           bool inorder, bool general);
int h2mi_msm_bn254_g1_phase_dev(uint64_t handle, const void* const* d_scalars, size_t count, size_t n, void* d_out_jacobian, unsigned flags,
                                h2mi_stream_t stream) {
  if (flags & ~(unsigned)(H2MI_MSM_SPARSE | H2MI_MSM_INORDER | H2MI_MSM_GENERAL)) return H2MI_EINVAL;
  return msm_batch_entry(handle, d_scalars, count, n, d_out_jacobian, stream, (flags & H2MI_MSM_SPARSE) && !ab_env("H2MI_MSM_IGNORE_SPARSE_HINT"),
                         (flags & H2MI_MSM_INORDER) && !ab_env("H2MI_MSM_IGNORE_INORDER"), (flags & H2MI_MSM_GENERAL) != 0);
}
static int msm_batch_entry(uint64_t handle, const void* const* d_scalars, size_t count, size_t n, void* d_out_jacobian, h2mi_stream_t stream, bool sparse,
                           bool inorder, bool general) {
  H2_REQUIRE_INIT();
  if (!d_out_jacobian || !d_scalars || count == 0) return H2MI_EINVAL;
  for (size_t j = 0; j < count; j++)
    if (!d_scalars[j] && n != 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  auto it = g_bases.find(handle);
  hipStream_t s = pick_stream(stream);
  const bool pipelined = (s == ctx().stream) && ctx().tail_stream && !getenv("H2MI_MSM_NO_PIPELINE");
  static const bool eager = ab_env("H2MI_MSM_EAGER_TAIL") != nullptr;
  // batched launches: a plain (unsharded) handle on the library stream, narrow windows or the small path, small enough that launch
  // pace and not stage overlap is what the phase waits for; everything else is the loop the caller would have written
  // `sparse` (the caller's promise that the columns are mostly zeros or one repeated value): the kernels of such an MSM are short at
  // EVERY size — a 2^20-row witness column is 25 us of digit counting and a handful of 6-us kernels — so the batch is taken at any size
  if (count == 1 && inorder) return msm_dev_entry(handle, d_scalars[0], n, d_out_jacobian, stream, true, general);
  if (n != 0 && count > 1 && !ab_env("H2MI_MSM_NO_HEAD_BATCH") && pipelined && !eager && it != g_bases.end() && n <= it->second->n &&
      (it->second->n <= (ab_env("H2MI_HEAD_BATCH_MAX_LOG") ? (size_t)1 << atoi(ab_env("H2MI_HEAD_BATCH_MAX_LOG")) : HEAD_BATCH_MAX_N) || sparse)) {
    Bases* B = it->second;
    if (B->small || B->seg_log == 0) {
      for (size_t j0 = 0; j0 < count;) {
        const size_t m = std::min({count - j0, (size_t)HEAD_BATCH, (size_t)B->nslot});
        const bool small = take_small(B, general);  // per group of launches: a long group crosses the streaming threshold on its way
        B->last_small = small;
        B->since_join += (uint32_t)m;
        int rc = small ? msm_small_batch(B, d_scalars + j0, m, n, (char*)d_out_jacobian + 96 * j0, s, inorder)
                       : msm_dev_batch(B, d_scalars + j0, m, n, (char*)d_out_jacobian + 96 * j0, s, inorder);
        if (rc) return rc;
        j0 += m;
      }
      return H2MI_OK;
    }
  }
  for (size_t j = 0; j < count; j++) {
    int rc = msm_dev_entry(handle, d_scalars[j], n, (char*)d_out_jacobian + 96 * j, stream, false, general);
    if (rc) return rc;
  }
  return H2MI_OK;
}

int h2mi_msm_bn254_g1(uint64_t handle, const uint64_t* bases, const uint64_t* scalars, size_t n, uint64_t out[12]) {
  H2_REQUIRE_INIT();
  if (!out) return H2MI_EINVAL;
  if (n == 0) {  // best_multiexp(&[], &[]) = identity
    memcpy(out, G1_IDENTITY, 96);
    return H2MI_OK;
  }
  if (!scalars) return H2MI_EINVAL;
  if (handle == 0 && !bases) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  uint64_t h = handle;
  int rc = H2MI_OK;
  auto shd = g_sharded.find(handle);
  if (shd != g_sharded.end()) {
    static uint8_t* d_res = nullptr;  // 96 B on the primary device
    if (!d_res) H2_HIP(hipMalloc((void**)&d_res, 96));
    hipStream_t s0 = ctx().devs[0].stream;
    rc = msm_sharded(shd->second, scalars, /*on_host=*/true, n, d_res);
    if (!rc) rc = msm_join_all(s0);
    if (!rc && hipMemcpyAsync(out, d_res, 96, hipMemcpyDeviceToHost, s0) != hipSuccess) rc = H2MI_EHIP;
    if (hipStreamSynchronize(s0) != hipSuccess && !rc) rc = H2MI_EHIP;
    return rc;
  }
  if (handle == 0) {
    rc = adhoc_handle(bases, n, &h);
    if (rc) return rc;
  }
  auto it = g_bases.find(h);
  if (it == g_bases.end()) return H2MI_EHANDLE;
  if (n > it->second->n) return H2MI_ERANGE;
  hipStream_t s = ctx().stream;
  Bases* B = it->second;
  // staging buffer kept with the handle: a prover calls commit() many times per proof, and a device
  // allocation + free of 32 MB per call costs as much as a tenth of the MSM itself
  if (!B->host_stage && hipMalloc(&B->host_stage, B->n * 32 + 96) != hipSuccess) {
    B->host_stage = nullptr;
    rc = H2MI_ENOMEM;
  }
  uint8_t* d = B->host_stage;
  if (!rc && hipMemcpyAsync(d + 96, scalars, n * 32, hipMemcpyHostToDevice, s) != hipSuccess) rc = H2MI_EHIP;
  // best_multiexp returns its point: the call is a lone MSM read back at once, so it runs in order on the library stream (no stream hops,
  // nothing deferred: ~50 us less than the pipelined form) and the 96 bytes come back through pinned memory
  if (!rc) rc = msm_dev(it->second, d + 96, n, d, s, /*inorder=*/!ab_env("H2MI_MSM_IGNORE_INORDER"));
  if (!rc) rc = msm_join_all(s);
  static thread_local void* pinned = nullptr;  // 96 B of pinned host memory per calling thread, kept for the life of the thread
  if (!pinned && hipHostMalloc(&pinned, 96, hipHostMallocPortable) != hipSuccess) pinned = nullptr;
  void* back = pinned ? pinned : (void*)out;
  if (!rc && hipMemcpyAsync(back, d, 96, hipMemcpyDeviceToHost, s) != hipSuccess) rc = H2MI_EHIP;
  if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = H2MI_EHIP;
  if (!rc && pinned) memcpy(out, pinned, 96);
  return rc;  // an ad-hoc registration (handle == 0) stays cached: see adhoc_handle
}

int h2mi_msm_adhoc_builds(uint64_t* builds_out) {
  if (!builds_out) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  *builds_out = g_adhoc_builds;
  return H2MI_OK;
}

int h2mi_msm_set_canonical(int on) {
  H2_REQUIRE_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  g_canonical = on != 0;
  return H2MI_OK;
}

int h2mi_msm_last_stats(uint64_t handle, uint64_t* bucket_adds, uint64_t* reduce_adds) {
  H2_REQUIRE_INIT();
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  auto sh = g_sharded.find(handle);
  if (sh != g_sharded.end()) {  // sums over the slices
    uint64_t ba = 0, ra = 0;
    for (Shard& sd : sh->second->shard) {
      uint64_t a = 0, r = 0;
      int rc = h2mi_msm_last_stats(sd.handle, &a, &r);
      if (rc) return rc;
      ba += a;
      ra += r;
    }
    if (bucket_adds) *bucket_adds = ba;
    if (reduce_adds) *reduce_adds = ra;
    return H2MI_OK;
  }
  auto it = g_bases.find(handle);
  if (it == g_bases.end()) return H2MI_EHANDLE;
  Bases* B = it->second;
  {
    int rc = flush_tails();
    if (rc) return rc;
  }
  H2_HIP(hipDeviceSynchronize());
  uint64_t st = 0;
  H2_HIP(hipMemcpy(&st, B->slot[B->last_slot].stats, 8, hipMemcpyDeviceToHost));
  if (bucket_adds) *bucket_adds = st;
  if (reduce_adds && B->last_small) {  // the quad trees: 255 additions per workgroup of 256 lanes, then over the partial sums
    *reduce_adds = (uint64_t)B->sG * (B->slanes - 1) + B->sG;
  } else if (reduce_adds) {
    // row + column tree sums touch every bucket twice; weighted sums and the final doublings are O(sqrt(nb))
    uint64_t Nh = 1ull << B->logNh, Nl = 1ull << B->logNl;
    const uint64_t mat = (uint64_t)B->nb >> B->seg_log;  // wide windows: 2 (L - 1) additions per segment of L buckets first
    *reduce_adds = (B->seg_log ? 2ull * (B->nb - mat) + mat : 0ull) + 2ull * mat + (B->logNh * Nh + (B->logNl + 1) * Nl) / 2 +
                   (B->logNh + B->logNl + 1) * (uint64_t)(B->c);
  }
  return H2MI_OK;
}

}  // extern "C"
