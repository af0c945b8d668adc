// Shared host/device helpers for libtgm_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tgm_amd.h"

namespace tgmx {

constexpr int kWave = 64;  // CDNA wavefront width

void set_error(const char* fmt, ...);

#define TGMX_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ::tgmx::set_error(__VA_ARGS__);    \
      return TGMX_E_INVALID;             \
    }                                    \
  } while (0)

#define TGMX_CHECK_LAUNCH(what)                                                        \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      ::tgmx::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));        \
      return TGMX_E_LAUNCH;                                                            \
    }                                                                                  \
  } while (0)

// q = n / d for n < 2^32 / d via one v_mul_hi_u32 (m = floor(2^32 / d) + 1).
struct FastDiv {
  uint32_t d, m, pass;  // pass: all ones when d <= 1 (m is 0 then)
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return d <= 1 ? n : __umulhi(n, m); }
  // the same without a branch (a uniform branch still ends the basic block: loads on either side of it are not batched)
  __device__ __forceinline__ uint32_t div_nb(uint32_t n) const { return __umulhi(n, m) + (n & pass); }
};
inline FastDiv make_fastdiv(uint32_t div) {
  FastDiv f;
  f.d = div;
  f.m = div <= 1 ? 0u : (uint32_t)((1ull << 32) / div) + 1u;
  f.pass = div <= 1 ? 0xFFFFFFFFu : 0u;
  return f;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// RandomNegativeEdgeSamplerHook's draw i of call `call` (tgm/hooks/negatives/sampler.py:45-65: uniform ids in [low, low + range)):
// a counter-based generator, so the stand-alone kernel (dedup.hip) and the seed fetch of the lookup kernels (recency.hip,
// negatives generated in place) produce the same ids
// counter-based generator: 32 uniform bits for (seed, stream, index) -- a splitmix64 finaliser over the mixed counter
__host__ __device__ __forceinline__ unsigned counter_u32(unsigned long long seed, unsigned long long stream, unsigned long long i) {
  unsigned long long x = seed ^ (stream * 0x9E3779B97F4A7C15ull) ^ (i * 0xD1B54A32D192ED03ull);
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (unsigned)(x >> 32);
}
__device__ __forceinline__ int negative_draw(unsigned long long seed, unsigned long long call, unsigned long long i, int low, unsigned range) {
  return low + (int)__umulhi(counter_u32(seed, call, i), range);  // uniform up to 2^-32 * range
}

// nn.Dropout(p) as a counter-based mask: element i of stream `stream` is kept iff its 32 random bits are >= thresh =
// floor(p * 2^32); kept elements are scaled by inv = 1 / (1 - p).  The backward passes regenerate the mask from the same
// (seed, stream, i) instead of storing it.  thresh == 0: dropout off (every element kept, scale 1).
struct DropoutArgs {
  unsigned thresh;
  float inv;
  unsigned long long seed, stream;
};
__device__ __forceinline__ float dropout_scale(const DropoutArgs& d, unsigned long long i) {
  if (d.thresh == 0) return 1.f;
  return counter_u32(d.seed, d.stream, i) >= d.thresh ? d.inv : 0.f;
}
inline DropoutArgs make_dropout(const tgmx_dropout_t* t) {
  DropoutArgs d{0u, 1.f, 0ull, 0ull};
  if (t && t->p > 0.f) {
    const double th = (double)t->p * 4294967296.0;
    d.thresh = th >= 4294967295.0 ? 4294967295u : (unsigned)th;
    d.inv = (float)(1.0 / (1.0 - (double)t->p));
    d.seed = t->seed;
    d.stream = t->stream;
  }
  return d;
}
inline DropoutArgs make_dropout(float p, unsigned long long seed, unsigned long long stream) {
  DropoutArgs d{0u, 1.f, seed, stream};
  if (p > 0.f) {
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    d.inv = (float)(1.0 / (1.0 - (double)p));
  }
  return d;
}

// cos(x) for Time2Vec arguments (float32 x up to ~2^31 * w).  A row's 100 frequencies span 9 decades, so the
// library cosf would take its large-argument (Payne-Hanek) path on every wave (~130 instructions).  Instead:
//   * polynomial: float minimax sin / cos on [-pi/4, pi/4], quadrant from the reduction's integer;
//   * reduction, |x| < 8e6 (k < 2^23): three float FMAs x - k*c1 - k*c2 - k*c3 with a three-term pi/2 -- the FMA
//     forms k*c exactly, so each step rounds once (<= 6e-8 absolute on a result of order 1);
//   * reduction, larger |x|: the same in double with a two-term pi/2 (exact to ~1e-16 * k for every float32 input).
// The choice is made per WAVE (__all), not per lane: no divergence, and at dataset scale the high-frequency columns
// sit in one half of the lanes' column pairs, so the other half always takes the cheap path.
// Max error < 2.5e-7 absolute.  Must be called from wave-uniform control flow.
__device__ __forceinline__ float cos_t2v_poly(float rf, int q) {
  const float r2 = rf * rf;
  float sp = -1.9515295891e-4f;
  sp = __fmaf_rn(sp, r2, 8.3321608736e-3f);
  sp = __fmaf_rn(sp, r2, -1.6666654611e-1f);
  const float sn = __fmaf_rn(sp * r2, rf, rf);
  float cp = 2.443315711809948e-5f;
  cp = __fmaf_rn(cp, r2, -1.388731625493765e-3f);
  cp = __fmaf_rn(cp, r2, 4.166664568298827e-2f);
  const float cs = __fmaf_rn(cp * r2, r2, __fmaf_rn(-0.5f, r2, 1.0f));
  const float v = (q & 1) ? sn : cs;  // q: 0 -> cos r, 1 -> -sin r, 2 -> -cos r, 3 -> sin r
  return (q == 1 || q == 2) ? -v : v;
}

// the two reductions on their own (callers that decide once for many arguments: the attention kernel)
__device__ __forceinline__ float cos_t2v_small(float x) {  // |x| < 8e6
  float k = __builtin_rintf(x * 0.636619772367581343f);
  float r = __fmaf_rn(-k, 1.57079637050628662109375f, x);
  r = __fmaf_rn(-k, -4.37113900018624283e-8f, r);
  // near 8e6 the float product x * (2/pi) is only good to ~half a unit, so k can be one off: one exact correction
  // step on the two-term remainder brings r back into [-pi/4, pi/4] (without it: 7.7e-6 worst error in [3e6, 8e6))
  const float adj = r > 0.78539819f ? 1.f : (r < -0.78539819f ? -1.f : 0.f);
  k += adj;
  r = __fmaf_rn(-adj, 1.57079637050628662109375f, r);
  r = __fmaf_rn(-adj, -4.37113900018624283e-8f, r);
  r = __fmaf_rn(-k, -1.71512449e-15f, r);
  return cos_t2v_poly(r, (int)k & 3);
}

__device__ __forceinline__ float cos_t2v_big(float x) {  // any float32
  const double xd = (double)x;
  const double kd = __builtin_rint(xd * 0.63661977236758134308);
  double r = __builtin_fma(-kd, 1.57079632679489655800e+00, xd);
  r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
  return cos_t2v_poly((float)r, (int)((long long)kd & 3));
}

constexpr float kCosSmallLimit = 8.0e6f;

__device__ __forceinline__ float cos_t2v(float x) {
  if (__all(fabsf(x) < kCosSmallLimit)) return cos_t2v_small(x);
  return cos_t2v_big(x);
}

__device__ __forceinline__ bool pair_after(long long ka, int pa, long long kb, int pb) {
  return ka > kb || (ka == kb && pa > pb);
}


// Bitonic network over P = blockDim.x (key, entry index) pairs, ONE per thread: distances < 64 are lane shuffles, the
// rest go through LDS.  PK: the entry index rides in the key's low 12 bits (see packed_key) -- one 64-bit compare and
// two shuffles per step instead of a pair compare and three.
constexpr int kPackBits = 12;  // entry index < 4096 = kBlockMaxM
constexpr long long kPackBias = 1ll << 50;

// int32-wrapped keys live in [-2^31 + tmin, 2^31 + tmax]: with |t| < 2^49 the biased key fits 51 bits
__device__ __forceinline__ bool can_pack(int key_wrap32, long long tmin, long long tmax) {
  return key_wrap32 != 0 && tmin > -(1ll << 49) && tmax < (1ll << 49);
}
__device__ __forceinline__ long long packed_key(long long key, int j) { return ((key + kPackBias) << kPackBits) | (long long)j; }

template <bool PK>
__device__ __forceinline__ void bitonic_sort_one(long long& key, int& pay, long long* s_key, int* s_pay, int tid, int P) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      long long pk;
      int pp = 0;
      if (jj < kWave) {
        pk = __shfl_xor(key, jj);
        if (!PK) pp = __shfl_xor(pay, jj);
      } else {
        s_key[tid] = key;
        if (!PK) s_pay[tid] = pay;
        __syncthreads();
        pk = s_key[tid ^ jj];
        if (!PK) pp = s_pay[tid ^ jj];
        __syncthreads();
      }
      const bool low = (tid & jj) == 0, asc = (tid & k) == 0;
      const bool mine_after = PK ? key > pk : pair_after(key, pay, pk, pp);
      if ((low == asc) == mine_after) {
        key = pk;
        if (!PK) pay = pp;
      }
    }
  }
  if (PK) pay = (int)(key & ((1 << kPackBits) - 1));
}


// cos AND sin of one argument from ONE range reduction (the attention backward needs both for every (slot, column) it covers: the
// cosine is the feature, the sine its derivative).  The cosine is cos_t2v's, bit for bit (same path decision, same reduction, same
// polynomial); the sine comes from the same reduced argument (error ~1e-7, like the cosine's).
__device__ __forceinline__ void sincos_t2v(float x, float& sn_out, float& cs_out) {
  float rf;
  int q;
  if (__all(fabsf(x) < kCosSmallLimit)) {
    float k = __builtin_rintf(x * 0.636619772367581343f);
    float r = __fmaf_rn(-k, 1.57079637050628662109375f, x);
    r = __fmaf_rn(-k, -4.37113900018624283e-8f, r);
    const float adj = r > 0.78539819f ? 1.f : (r < -0.78539819f ? -1.f : 0.f);
    k += adj;
    r = __fmaf_rn(-adj, 1.57079637050628662109375f, r);
    r = __fmaf_rn(-adj, -4.37113900018624283e-8f, r);
    r = __fmaf_rn(-k, -1.71512449e-15f, r);
    rf = r;
    q = (int)k & 3;
  } else {
    const double xd = (double)x;
    const double kd = __builtin_rint(xd * 0.63661977236758134308);
    double r = __builtin_fma(-kd, 1.57079632679489655800e+00, xd);
    r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
    rf = (float)r;
    q = (int)((long long)kd & 3);
  }
  const float r2 = rf * rf;
  float sp = -1.9515295891e-4f;
  sp = __fmaf_rn(sp, r2, 8.3321608736e-3f);
  sp = __fmaf_rn(sp, r2, -1.6666654611e-1f);
  const float sn = __fmaf_rn(sp * r2, rf, rf);
  float cp = 2.443315711809948e-5f;
  cp = __fmaf_rn(cp, r2, -1.388731625493765e-3f);
  cp = __fmaf_rn(cp, r2, 4.166664568298827e-2f);
  const float cs = __fmaf_rn(cp * r2, r2, __fmaf_rn(-0.5f, r2, 1.0f));
  // q: 0 -> (sin r, cos r), 1 -> (cos r, -sin r), 2 -> (-sin r, -cos r), 3 -> (-cos r, sin r)
  const float cv = (q & 1) ? sn : cs;
  cs_out = (q == 1 || q == 2) ? -cv : cv;
  const float sv = (q & 1) ? cs : sn;
  sn_out = (q >= 2) ? -sv : sv;
}

// sin counterpart of cos_t2v (backward passes only; always the double-precision range reduction)
__device__ __forceinline__ float sin_t2v(float x) {
  const double xd = (double)x;
  const double kd = __builtin_rint(xd * 0.63661977236758134308);
  double r = __builtin_fma(-kd, 1.57079632679489655800e+00, xd);
  r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
  const float rf = (float)r;
  const int q = (int)((long long)kd & 3);
  const float r2 = rf * rf;
  float sp = -1.9515295891e-4f;
  sp = __fmaf_rn(sp, r2, 8.3321608736e-3f);
  sp = __fmaf_rn(sp, r2, -1.6666654611e-1f);
  const float sn = __fmaf_rn(sp * r2, rf, rf);
  float cp = 2.443315711809948e-5f;
  cp = __fmaf_rn(cp, r2, -1.388731625493765e-3f);
  cp = __fmaf_rn(cp, r2, 4.166664568298827e-2f);
  const float cs = __fmaf_rn(cp * r2, r2, __fmaf_rn(-0.5f, r2, 1.0f));
  const float v = (q & 1) ? cs : sn;  // q: 0 -> sin r, 1 -> cos r, 2 -> -sin r, 3 -> -cos r
  return (q >= 2) ? -v : v;
}

// ---- the sampled edge list's row scan (examples/linkproppred/tgn.py:80-92; csrc/tgn.hip) as a device function: ONE workgroup (any size that
// is a multiple of 64, <= 1024) counts a seed row's valid slots and scans the counts.  Its own launch in tgmx_tgn_edge_list; in the lowered
// chain it RIDES the unique-ids marking launch as one more workgroup (it reads the sampler's outputs only), so the list's write launch
// finds the offsets ready -- one single-workgroup launch (6 us of pure latency per batch) less on the critical path.
__device__ __forceinline__ void edge_list_scan_body(const int32_t* __restrict__ nbr, long long S, int k, int64_t* __restrict__ row_off,
                                                    int64_t* __restrict__ count) {
  __shared__ long long els_wave_tot[16];
  __shared__ long long els_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, P = blockDim.x;
  if (tid == 0) els_carry = 0;
  __syncthreads();
  for (long long r0 = 0; r0 < S; r0 += P) {
    const long long r = r0 + tid;
    long long c = 0;
    if (r < S)
      for (int s = 0; s < k; ++s) c += nbr[r * k + s] != -1;
    long long incl = c;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const long long o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) els_wave_tot[wave] = incl;
    __syncthreads();
    long long before = els_carry;
    for (int w = 0; w < wave; ++w) before += els_wave_tot[w];
    if (r < S) row_off[r] = before + incl - c;
    __syncthreads();
    if (tid == P - 1) els_carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) {
    row_off[S] = els_carry;
    *count = els_carry;
  }
}
struct EdgeListScan {  // the rider's arguments (nbr == NULL: none)
  const int32_t* nbr;
  long long S;
  int k;
  int64_t* row_off;
  int64_t* count;
};
}  // namespace tgmx
// the lowered chain's entries (csrc/pipeline.hip): the unique ids with the edge list's scan riding along, and the list with its scan done
int tgmx_internal_unique_ids(const int32_t* const* parts, const int64_t* part_sizes, int32_t num_parts, int32_t num_nodes, void* workspace,
                             int32_t* out_ids, int64_t* out_count, int32_t* status, const tgmx::EdgeListScan* rider, tgmx_stream_t stream);
int tgmx_internal_edge_list(const int32_t* seed, const int32_t* nbr, const int64_t* nbr_t, const float* nbr_x, const int32_t* nbr_eid, const float* table,
                            int64_t S, int32_t k, int32_t D, const int32_t* uniq, int64_t U, const int64_t* uniq_count, int64_t cap, int64_t* row_off,
                            int64_t* edge_index, int64_t* edge_t, float* edge_x, int64_t* count, bool scan_done, tgmx_stream_t stream);
namespace tgmx {
// one C = act(A B^T + bias) problem of tgmx_sgemm_nt's argument list (tgmx_internal_sgemm_nt_pair, csrc/tgat.hip)
struct GemmCall {
  const float* A;
  long long lda;
  const float* B;
  long long ldb;
  float* C;
  long long ldc, M;
  int N, K;
  const float* bias;
  int relu, batch;
  long long sA, sB, sC;
};
}  // namespace tgmx
// two independent GEMMs as one launch (bit for bit what two tgmx_sgemm_nt calls give; falls back to them when either problem would not
// take the K-split kernel on its own)
int tgmx_internal_sgemm_nt_pair(const tgmx::GemmCall& c0, const tgmx::GemmCall& c1, tgmx_stream_t stream);
