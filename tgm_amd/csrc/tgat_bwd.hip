// TGAT backward building blocks (fp32) for gfx950 -- the training path of the folded attention
// (forward: tgat.hip; mathematics: oracle/tgat_fold.py, differentiated by hand).
//
//   sgemm_tn        dW[M,N] = sum_r A[r,m] B[r,n]   weight gradients: reduction over the ROW index, both
//                   operands read row-major (coalesced along m / n), exact-fp32 MFMA, rows split over
//                   blocks into partials that a second kernel sums in a fixed order (deterministic)
//   colsum          bias / LayerNorm / Time2Vec parameter gradients (same two-stage scheme)
//   ln_backward     per-row LayerNorm backward (one wave per row)
//   attn_backward   per-row backward of masked-softmax attention over the k slots (one wave per row):
//                   d(folded query), d(neighbor features), per-row Time2Vec gradient partials
// The "NN" products (dX = dY W) reuse sgemm_nt with transposed weight copies.
#include "common.h"
#include "lanes.h"

#include <type_traits>

namespace tgmx {

using floatx16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

// ---------------------------------------------------------------------------
// partial[z][m][n] = sum over rows r in split z of A[r, m] * B[r, n]
// ---------------------------------------------------------------------------
struct GemmTnArgs {
  const float* A;
  const float* B;
  float* partial;  // [batch, splits, M, N]
  long long lda, ldb, sA, sB;
  long long R, rows_per_split;
  int M, N, splits;
};

// workgroup (bx, by, bz) of the grid [ceil(M / 128), ceil(N / 64), batch * splits]
__device__ __forceinline__ void sgemm_tn_block(const GemmTnArgs& g, int bx, int by, int bz) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int batch = bz / g.splits, split = bz - batch * g.splits;
  const int m0 = (bx * 4 + wave) * 32;
  if (m0 >= g.M) return;
  const int n0 = by * 64;
  const float* __restrict__ A = g.A + (long long)batch * g.sA;
  const float* __restrict__ B = g.B + (long long)batch * g.sB;
  const long long r0 = (long long)split * g.rows_per_split;
  long long r1 = r0 + g.rows_per_split;
  if (r1 > g.R) r1 = g.R;
  // columns past the edge are clamped for the loads; their results are never stored
  const int ma = m0 + i < g.M ? m0 + i : g.M - 1;
  const int nb0 = n0 + i < g.N ? n0 + i : g.N - 1;
  const int nb1 = n0 + 32 + i < g.N ? n0 + 32 + i : g.N - 1;
  floatx16 acc0 = {0}, acc1 = {0};
  // 4 MFMA steps (k = 2 rows each) per iteration; the NEXT iteration's twelve values are requested before this one's MFMAs (the loop was
  // load -> wait -> 8 MFMAs: one exposed latency per 8 rows; same values in the same order: bit-identical)
  float a[4], b0[4], b1[4], an[4], b0n[4], b1n[4];
  auto fetch = [&](long long r, float (&x)[4], float (&y0)[4], float (&y1)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long rr = r + 2 * u + half;
      const bool ok = rr < r1;
      const long long rc = ok ? rr : r0;
      const float av = A[rc * g.lda + ma], bv0 = B[rc * g.ldb + nb0], bv1 = B[rc * g.ldb + nb1];
      x[u] = ok ? av : 0.f;
      y0[u] = ok ? bv0 : 0.f;
      y1[u] = ok ? bv1 : 0.f;
    }
  };
  if (r0 < r1) fetch(r0, a, b0, b1);
  for (long long r = r0; r < r1; r += 8) {
    const bool more = r + 8 < r1;
    if (more) fetch(r + 8, an, b0n, b1n);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b0[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b1[u], acc1, 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = an[u];
        b0[u] = b0n[u];
        b1[u] = b1n[u];
      }
    }
  }
  float* __restrict__ P = g.partial + ((long long)batch * g.splits + split) * g.M * g.N;
#pragma unroll
  for (int rg = 0; rg < 16; ++rg) {
    const int row = m0 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
    if (row >= g.M) continue;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = n0 + t * 32 + i;
      if (col < g.N) P[(long long)row * g.N + col] = t ? acc1[rg] : acc0[rg];
    }
  }
}

__global__ __launch_bounds__(256) void sgemm_tn_kernel(const GemmTnArgs g) { sgemm_tn_block(g, blockIdx.x, blockIdx.y, blockIdx.z); }

// several independent A^T B products as ONE launch (a layer's weight gradients: each alone is 500-700 workgroups of one
// dependent MFMA chain per wave -- 20-30 us of latency on a third of the chip; together they fill it)
constexpr int kTnJobs = 16;
struct GemmTnJobs {
  GemmTnArgs job[kTnJobs];
  int first_block[kTnJobs + 1];  // prefix sums of the jobs' workgroup counts
  int gx[kTnJobs], gy[kTnJobs];
  int n;
};
__global__ __launch_bounds__(256) void sgemm_tn_batch_kernel(const GemmTnJobs J) {
  int j = 0;
  while (j + 1 < J.n && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
  const int local = (int)blockIdx.x - J.first_block[j];
  const int bx = local % J.gx[j], rest = local / J.gx[j];
  sgemm_tn_block(J.job[j], bx, rest % J.gy[j], rest / J.gy[j]);
}

// out[b][m * ldo + n] (+)= sum_s partial[b][s][m][n]   (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int splits, int M, int N,
                                                              float* __restrict__ out, long long ldo, long long s_out, int accumulate) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long MN = (long long)M * N;
  if (e >= MN) return;
  const float* p = partial + (long long)blockIdx.y * splits * MN + e;
  float acc = 0.f;
  for (int s = 0; s < splits; s += 8) {  // 8 independent loads in flight; the summation order stays 0, 1, 2, ...
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s + u < splits ? p[(long long)(s + u) * MN] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  const int m = (int)(e / N), n = (int)(e - (long long)m * N);
  float* o = out + (long long)blockIdx.y * s_out + (long long)m * ldo + n;
  *o = accumulate ? *o + acc : acc;
}

// partial[z][c] = sum over rows of split z of in[r * ld + c] * (scale ? scale[c] : 1)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ in, long long ld, long long R, int C,
                                                             long long rows_per_split, float* __restrict__ partial) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long long r0 = (long long)blockIdx.y * rows_per_split;
  long long r1 = r0 + rows_per_split;
  if (r1 > R) r1 = R;
  float acc = 0.f;
  for (long long r = r0; r < r1; r += 8) {  // 8 independent loads in flight; summation order stays r0, r0+1, ...
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = r + u < r1 ? in[(r + u) * ld + c] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  partial[(long long)blockIdx.y * C + c] = acc;
}

// g[r, c] = h[r, c] > 0 ? g[r, c] : 0
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ g, long long ldg, const float* __restrict__ h, long long ldh,
                                                        long long R, int C) {
  const long long total = R * C;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long r = e / C;
    const int c = (int)(e - r * C);
    if (!(h[r * ldh + c] > 0.f)) g[r * ldg + c] = 0.f;
  }
}

// dst[r, c] (+)= src[r, c] for c < C
__global__ __launch_bounds__(256) void add_cols_kernel(float* __restrict__ dst, long long ldd, const float* __restrict__ src, long long lds,
                                                       long long R, int C, int accumulate) {
  const long long total = R * C;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long r = e / C;
    const int c = (int)(e - r * C);
    const float v = src[r * lds + c];
    dst[r * ldd + c] = accumulate ? dst[r * ldd + c] + v : v;
  }
}

// LayerNorm backward, one wave per row: u = y + res, xhat = (u - mean) * rstd,
//   du = rstd * (g*dout - mean(g*dout) - xhat * mean(g*dout*xhat)),  dgx = dout * xhat
__global__ __launch_bounds__(256) void ln_backward_kernel(const float* __restrict__ dout, long long ldd, const float* __restrict__ y,
                                                          long long ldy, const float* __restrict__ res, long long ldr,
                                                          const float* __restrict__ gamma, int O, float eps, long long R,
                                                          float* __restrict__ du, long long ldu, float* __restrict__ dgx, long long ldg) {
  const int lane = lane_id();
  const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* yr = y + r * ldy;
  const float* rr = res + r * ldr;
  const float* dr = dout + r * ldd;
  float s = 0.f;
  for (int c = lane; c < O; c += kWave) s += yr[c] + rr[c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)O;
  float v = 0.f;
  for (int c = lane; c < O; c += kWave) {
    const float t = yr[c] + rr[c] - mean;
    v += t * t;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = 1.0f / sqrtf(v / (float)O + eps);
  float a1 = 0.f, a2 = 0.f;
  for (int c = lane; c < O; c += kWave) {
    const float xh = (yr[c] + rr[c] - mean) * rstd;
    const float gd = gamma[c] * dr[c];
    a1 += gd;
    a2 += gd * xh;
  }
  for (int o = 32; o > 0; o >>= 1) {
    a1 += __shfl_xor(a1, o);
    a2 += __shfl_xor(a2, o);
  }
  a1 /= (float)O;
  a2 /= (float)O;
  for (int c = lane; c < O; c += kWave) {
    const float xh = (yr[c] + rr[c] - mean) * rstd;
    du[r * ldu + c] = rstd * (gamma[c] * dr[c] - a1 - xh * a2);
    dgx[r * ldg + c] = dr[c] * xh;
  }
}

// ---------------------------------------------------------------------------
// attention backward, one wave per row (k * H <= 64)
// ---------------------------------------------------------------------------
struct AttnBwdArgs {
  const float* qf;      // [R, H, Cs]
  const float* probs;   // [R, H, k]
  const float* dzbar;   // [R, H, Cs]
  const float* nbrf;    // [R, k, d]
  const float* ex;      // [R, k, D]; or NULL with the two below: edge features by id (tgmx_tgat_hop_t.nbr_eid / edge_table)
  const int32_t* eid;   // [R, k] edge id behind every slot (-1: a pad slot, zeros)
  const float* table;   // [E, D] the resident store's feature rows
  const int64_t* seed_t;
  const int64_t* nbr_t;
  const float* tw;
  const float* tb;
  float* dqf;           // [R, H, Cs]
  float* dnbr;          // [R, k, d] accumulated (+=), may be null
  float* dtime;         // [R, 2T]: per-row partials of d(tw) | d(tb)
  long long R;
  int d, D, T, k, C, Cs;
  float scale;
  DropoutArgs drop;  // the forward's attention dropout: regenerated, element ((row0 + r) * H + h) * k + s
  long long drop_row0;
  // Several levels of the hop tree in ONE launch (tgmx_tgat_backward: they share the layer's weights; qf / probs / dzbar / dqf /
  // dtime are contiguous over the layer's rows, indexed by the layer-wide row r): level i covers rows [seg_begin[i], seg_begin[i+1])
  // and brings its own sampler tensors, biased by the caller so that the layer-wide r addresses them.  n_seg = 0: the fields above.
  int n_seg;
  long long seg_begin[TGMX_TGAT_MAX_LAYERS];
  const float* seg_nbrf[TGMX_TGAT_MAX_LAYERS];
  const float* seg_ex[TGMX_TGAT_MAX_LAYERS];
  const int32_t* seg_eid[TGMX_TGAT_MAX_LAYERS];
  const float* seg_table[TGMX_TGAT_MAX_LAYERS];
  const int64_t* seg_seed_t[TGMX_TGAT_MAX_LAYERS];
  const int64_t* seg_nbr_t[TGMX_TGAT_MAX_LAYERS];
  float* seg_dnbr[TGMX_TGAT_MAX_LAYERS];
};

template <int HALF>
__device__ __forceinline__ void bwd_reduce_scatter_step(float (&P)[64], int lane) {
  const bool upper = (lane & HALF) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const float send = upper ? P[i] : P[i + HALF];
    const float recv = __shfl_xor(send, HALF);
    const float keep = upper ? P[i + HALF] : P[i];
    P[i] = keep + recv;
  }
}

// The same reduce-scatter over the first NVp (a power of two) entries only: the steps whose stride is >= NVp cannot scatter -- both
// partners add (a + b == b + a: the value the full version leaves in the lane that keeps the entry) -- the rest are the steps above.
// Lane j ends up with entry j mod NVp, summed over the 64 lanes in the order xor 32, 16, 8, 4, 2, 1 whatever NVp is.
template <int NVp, int HALF>
__device__ __forceinline__ void bwd_reduce_n_step(float (&P)[NVp], int lane) {
  if constexpr (HALF >= NVp) {
#pragma unroll
    for (int i = 0; i < NVp; ++i) P[i] += __shfl_xor(P[i], HALF);
  } else {
    const bool upper = (lane & HALF) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const float send = upper ? P[i] : P[i + HALF];
      const float recv = __shfl_xor(send, HALF);
      const float keep = upper ? P[i + HALF] : P[i];
      P[i] = keep + recv;
    }
  }
}
template <int NVp>
__device__ __forceinline__ void bwd_reduce_scatter_n(float (&P)[NVp], int lane) {
  bwd_reduce_n_step<NVp, 32>(P, lane);
  bwd_reduce_n_step<NVp, 16>(P, lane);
  bwd_reduce_n_step<NVp, 8>(P, lane);
  bwd_reduce_n_step<NVp, 4>(P, lane);
  bwd_reduce_n_step<NVp, 2>(P, lane);
  bwd_reduce_n_step<NVp, 1>(P, lane);
}

// KMAX: the most slots a row of this launch can have (k <= KMAX): the widest body instantiated -- the register allocation of the kernel
// is its widest body's, and k <= 20 (every example configuration) should not pay for a 32-slot one
template <int H, int KMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void tgat_attn_backward_kernel(const AttnBwdArgs a) {
  constexpr int G = (64 / H) < KMAX ? (64 / H) : KMAX;
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int k = a.k, T = a.T, d = a.d, D = a.D, Cs = a.Cs;
  // (cosines AND sines are staged, one range reduction for each pair: 16.5 KB per wave at k = 20, T = 100 -- affordable since a workgroup
  // is one wave; the sines used to be re-evaluated where the time gradient consumes them, through the double-precision reduction: a
  // third of the 2 800 vector instructions a row of this kernel executed, profiles/r05_attention_counters.md)
  float* s_cos = lds_all + (size_t)wave * (2 * k * T + 2 * k + 2 * H * k);
  float* s_sin = s_cos + k * T;  // the sines of the same arguments (one range reduction for both: sincos_t2v)
  float* s_dt = s_sin + k * T;
  float* s_A = s_dt + k;
  float* s_ds = s_A + H * k;
  int* s_eid = reinterpret_cast<int*>(s_ds + H * k);  // [k] (edge features by id)
  const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  if (r >= a.R) return;
  const float* __restrict__ q = a.qf + r * (long long)H * Cs;
  const float* __restrict__ dz = a.dzbar + r * (long long)H * Cs;
  // this row's level (wave-uniform) and its tensors
  const float* lv_nbrf = a.nbrf;
  const float* lv_ex = a.ex;
  const int32_t* lv_eid = a.eid;
  const float* lv_table = a.table;
  const int64_t *lv_seed_t = a.seed_t, *lv_nbr_t = a.nbr_t;
  float* lv_dnbr = a.dnbr;
  if (a.n_seg > 0) {
    int sg = 0;
    for (int i2 = 1; i2 < a.n_seg; ++i2)
      if (r >= a.seg_begin[i2]) sg = i2;
    lv_nbrf = a.seg_nbrf[sg]; lv_ex = a.seg_ex[sg]; lv_eid = a.seg_eid[sg]; lv_table = a.seg_table[sg];
    lv_seed_t = a.seg_seed_t[sg]; lv_nbr_t = a.seg_nbr_t[sg]; lv_dnbr = a.seg_dnbr[sg];
  }
  const float* __restrict__ nb = lv_nbrf + r * (long long)k * d;
  const float* __restrict__ ex = lv_ex ? lv_ex + r * (long long)k * D : nullptr;
  float* __restrict__ dq = a.dqf + r * (long long)H * Cs;

  // ---- rows without upstream gradient: nothing to compute ----
  // A level-1 row that layer 2 consumed through a MASKED slot (a pad seed: ~1/4 of the layer-1 rows at the headline shape) receives
  // dzs = A' dzbar + scale ds q = 0 exactly, and zero rows stay zero through the tail's backward (GEMMs, LayerNorm, ReLU mask): its dzbar
  // is all zeros, so dA = 0, ds = 0 and every output of this row is 0 -- written as such, without reading a feature.
  {
    bool nz = false;
    for (int c = lane; c < a.C; c += kWave) {
#pragma unroll
      for (int h = 0; h < H; ++h) nz = nz || dz[h * Cs + c] != 0.f;
    }
    if (!__any(nz)) {
      for (int c = lane; c < H * Cs; c += kWave) dq[c] = 0.f;
      for (int c = lane; c < 2 * T; c += kWave) a.dtime[r * 2LL * T + c] = 0.f;
      return;
    }
  }
  // ---- the slots that carry gradient ----
  // lane j = s * H + h holds (slot s, head h).  A masked slot of a row that has a valid one has A == 0 in every head (exp(-1e10 - max)
  // underflows): its ds = A (dA - dot) is 0 whatever dA is, and with it every contribution of the slot to dqf, dnbr and the Time2Vec
  // gradient.  The sampler right-aligns a window (pads on the left; ~7 of 20 slots of a real row hold a neighbor at the headline
  // shape), so the slots LEFT of the first one with A != 0 are not read, scored or differentiated at all: the body below is
  // instantiated for the last GS = 4, 8, 12, 16, 20, 24 and all G slot positions and a row takes the smallest that covers its span
  // (straight-line code per body: guards that skip single slots serialise the loads -- measured, 241 -> 340 us).  A dead slot
  // INSIDE the span (an interior pad, an explicit mask) is computed like before: A = 0 makes its contributions exact zeros.  A row with
  // no valid slot attends uniformly (A = 1/k): its span is all k slots.  Every sum keeps its order: results equal the all-slots body's.
  const long long st = lv_seed_t[r];
  for (int s = lane; s < k; s += kWave) {
    s_dt[s] = (float)(st - lv_nbr_t[r * k + s]);
    if (lv_eid) s_eid[s] = lv_eid[r * k + s];
  }
  int span;
  {
    const int js0 = lane / H, jh0 = lane - js0 * H;
    const float A0 = js0 < k ? a.probs[r * (long long)H * k + jh0 * k + js0] : 0.f;
    const unsigned long long amask = __ballot(A0 != 0.f);
    const int first = amask ? (__ffsll((long long)amask) - 1) / H : 0;
    span = k - first;
  }
  __builtin_amdgcn_wave_barrier();
  // feature c of slot sl: a strided block of the row's own slots, or (edge features by id) row eid[sl] of the resident store --
  // the id is the same in every lane (a scalar base for the load); a pad slot (-1) reads zeros like the dense copy holds them
  auto dense = [](const float* __restrict__ base, long long slot_stride) {
    return [=](int sl, int c) -> float { return base[(long long)sl * slot_stride + c]; };
  };
  auto by_id = [&](int sl, int c) -> float {
    const int e = __builtin_amdgcn_readfirstlane(s_eid[sl]);
    return e >= 0 ? lv_table[(long long)e * D + c] : 0.f;
  };
  auto body = [&](auto gsc) __attribute__((always_inline)) {
    constexpr int GS = decltype(gsc)::value;  // slot positions this body covers: slots [k - GS, k), those below 0 do not exist
    const int sb = k - GS;
    for (int i = 0; i < GS; ++i) {  // the cosines of the covered slots
      const int sl = sb + i;
      if (sl < 0) continue;
      const float dts = s_dt[sl];
      for (int t = lane; t < T; t += kWave) {
        float sn, cs;
        sincos_t2v(__fmaf_rn(dts, a.tw[t], a.tb[t]), sn, cs);
        s_cos[sl * T + t] = cs;
        s_sin[sl * T + t] = sn;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- dA[h][s] = dzbar[h] . z[s] ----
    // (position, head) pairs 0 .. 31 in one reduce-scatter of <= 32 entries, the rest (k = 20, two heads: 8 more) in a second, small one:
    // 40 registers of partial sums instead of 64 -- the body must stay under the two-waves-per-SIMD register budget
    constexpr int NV = GS * H, NV1 = NV < 32 ? NV : 32, NV2 = NV - NV1;
    constexpr int NVp = NV1 <= 2 ? 2 : NV1 <= 4 ? 4 : NV1 <= 8 ? 8 : NV1 <= 16 ? 16 : 32;
    constexpr int NVq = NV2 <= 0 ? 1 : NV2 <= 2 ? 2 : NV2 <= 4 ? 4 : NV2 <= 8 ? 8 : NV2 <= 16 ? 16 : 32;
    float P[NVp], Q[NVq];
#pragma unroll
    for (int j = 0; j < NVp; ++j) P[j] = 0.f;
#pragma unroll
    for (int j = 0; j < NVq; ++j) Q[j] = 0.f;
    // A section of `dim` columns (neighbor features, edge features, time encoding) takes ceil(dim / 64) column chunks per lane.  Up to
    // three chunks (dim <= 192) are ONE straight-line block: the chunks' loads are all issued before the first product -- a chunk loop
    // is a dependent round trip per chunk (a row of this kernel was a chain of ~12 of them: 45 us of wave life per row).  A column past
    // the section reads column 0 with a zero multiplier.  Per accumulator the products still arrive in chunk order.
    auto accumulate = [&](auto&& load, int dim, int col0) {
      auto run = [&](auto nch) __attribute__((always_inline)) {
        constexpr int NCH = decltype(nch)::value;
        float gv[NCH][H];
        int cc[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
          const int c = lane + u * kWave;
          const bool ok = c < dim;
          cc[u] = ok ? c : 0;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const float g = dz[h * Cs + col0 + cc[u]];
            gv[u][h] = ok ? g : 0.f;
          }
        }
#pragma clang loop unroll(full)
        for (int i = 0; i < GS; ++i) {
          const int sl = sb + i > 0 ? sb + i : 0;  // (a position below slot 0 re-reads slot 0: its lanes are not live)
#pragma unroll
          for (int u = 0; u < NCH; ++u) {
            const float z = load(sl, cc[u]);
#pragma unroll
            for (int h = 0; h < H; ++h) {
              if (i * H + h < 32) P[(i * H + h) & (NVp - 1)] = __fmaf_rn(gv[u][h], z, P[(i * H + h) & (NVp - 1)]);  // (compile-time indices)
              else Q[(i * H + h - 32) & (NVq - 1)] = __fmaf_rn(gv[u][h], z, Q[(i * H + h - 32) & (NVq - 1)]);
            }
          }
        }
      };
      if (dim <= kWave) run(std::integral_constant<int, 1>{});
      else if (dim <= 2 * kWave) run(std::integral_constant<int, 2>{});
      else if (dim <= 3 * kWave) run(std::integral_constant<int, 3>{});
      else {
        for (int c0 = 0; c0 < dim; c0 += kWave) {  // wide sections: chunk by chunk
          const int c = c0 + lane;
          const bool ok = c < dim;
          const int ce = ok ? c : 0;
          float gv[H];
#pragma unroll
          for (int h = 0; h < H; ++h) gv[h] = ok ? dz[h * Cs + col0 + ce] : 0.f;
#pragma clang loop unroll(full)
          for (int i = 0; i < GS; ++i) {
            const float z = load(sb + i > 0 ? sb + i : 0, ce);
#pragma unroll
            for (int h = 0; h < H; ++h) {
              if (i * H + h < 32) P[(i * H + h) & (NVp - 1)] = __fmaf_rn(gv[h], z, P[(i * H + h) & (NVp - 1)]);
              else Q[(i * H + h - 32) & (NVq - 1)] = __fmaf_rn(gv[h], z, Q[(i * H + h - 32) & (NVq - 1)]);
            }
          }
        }
      }
    };
    accumulate(dense(nb, d), d, 0);
    if (D > 0) {
      if (ex) accumulate(dense(ex, D), D, d);
      else accumulate(by_id, D, d);
    }
    accumulate(dense(s_cos, T), T, d + D);
    bwd_reduce_scatter_n<NVp>(P, lane);
    if constexpr (NV2 > 0) bwd_reduce_scatter_n<NVq>(Q, lane);  // lane 32 + j (j < NV2 <= NVq) ends up with entry 32 + j
    const int ji = lane / H, jh = lane - ji * H;  // lane j = i * H + h (j < NV) holds position i of the body, head h
    const int js = sb + ji;
    const bool live = ji < GS && js >= 0;
    // zbar used A' = A * mk (dropout on the softmax output, attention.py:119): dA = dA' * mk, and the slot gradients below take A'
    const float mk = live ? dropout_scale(a.drop, (unsigned long long)(a.drop_row0 + r) * (H * k) + jh * k + js) : 0.f;
    const float dA = ((NV2 > 0 && lane >= 32) ? Q[0] : P[0]) * mk;
    const float A = live ? a.probs[r * (long long)H * k + jh * k + js] : 0.f;
    float dot = A * dA;
#pragma unroll
    for (int o = H; o < 64; o <<= 1) dot += __shfl_xor(dot, o);
    const float ds = A * (dA - dot);  // softmax backward; masked slots have A == 0
    if (live) {
      s_A[jh * k + js] = A * mk;
      s_ds[jh * k + js] = ds;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- dqf[h][c] = scale * sum_s ds[h][s] z[s][c];  dz[s][c] = sum_h A dzbar + scale * ds * qf ----
    // one column of one section: the covered slots' values of that column (all loaded first), then the slot loop
    auto column = [&](auto&& load, int c, bool ok, int col0, auto partc) __attribute__((always_inline)) {
      constexpr int part = decltype(partc)::value;  // 0: neighbor features (gradient flows on: dnbr), 1: edge features (none), 2: time encoding
      const int ce = ok ? c : 0;
      float acc[H], gv[H], qv[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        acc[h] = 0.f;
        gv[h] = dz[h * Cs + col0 + ce];
        qv[h] = q[h * Cs + col0 + ce];
      }
      float dw = 0.f, db = 0.f;
      float zs[GS];
#pragma clang loop unroll(full)
      for (int i = 0; i < GS; ++i) zs[i] = load(sb + i > 0 ? sb + i : 0, ce);
      // dnbr is ACCUMULATED into: its old values are read here, all at once -- `dnbr[...] += dzs` inside the slot loop is a load the
      // compiler must order behind the previous slot's store (it cannot prove the rows distinct): GS dependent round trips per column
      // chunk, 60 per row of a 172-feature layer -- most of that launch's 53 us
      float dn[part == 0 ? GS : 1];
      const bool dn_on = part == 0 && lv_dnbr != nullptr;
      if constexpr (part == 0) {
#pragma clang loop unroll(full)
        for (int i = 0; i < GS; ++i) dn[i] = dn_on ? lv_dnbr[(r * k + (sb + i > 0 ? sb + i : 0)) * (long long)d + ce] : 0.f;
      }
      if (!ok) return;  // (a lane without a column in this chunk: its loads above keep the chunk one block of loads; nothing to store)
#pragma clang loop unroll(full)
      for (int i = 0; i < GS; ++i) {
        const int sl = sb + i;
        if (sl >= 0) {  // (wave-uniform)
          const float z = zs[i];
          float dzs = 0.f;
#pragma unroll
          for (int h = 0; h < H; ++h) {
            acc[h] = __fmaf_rn(s_ds[h * k + sl], z, acc[h]);
            dzs += s_A[h * k + sl] * gv[h] + a.scale * s_ds[h * k + sl] * qv[h];
          }
          if constexpr (part == 0) dn[i] += dzs;
          if constexpr (part == 2) {
            const float g = -s_sin[sl * T + c] * dzs;  // d cos(arg) / d arg: the sine staged with the cosine (same argument, same reduction)
            dw = __fmaf_rn(g, s_dt[sl], dw);
            db += g;
          }
        }
      }
      if constexpr (part == 0) {
        if (dn_on) {
#pragma clang loop unroll(full)
          for (int i = 0; i < GS; ++i)
            if (sb + i >= 0) lv_dnbr[(r * k + sb + i) * (long long)d + c] = dn[i];
        }
      }
#pragma unroll
      for (int h = 0; h < H; ++h) dq[h * Cs + col0 + c] = acc[h] * a.scale;
      if constexpr (part == 2) {
        a.dtime[r * 2LL * T + c] = dw;
        a.dtime[r * 2LL * T + T + c] = db;
      }
    };
    auto columns = [&](auto&& load, int dim, int col0, auto partc) {
      for (int c0 = 0; c0 < dim; c0 += kWave) column(load, c0 + lane, c0 + lane < dim, col0, partc);
    };
    columns(dense(nb, d), d, 0, std::integral_constant<int, 0>{});
    if (D > 0) {
      if (ex) columns(dense(ex, D), D, d, std::integral_constant<int, 1>{});
      else columns(by_id, D, d, std::integral_constant<int, 1>{});
    }
    columns(dense(s_cos, T), T, d + D, std::integral_constant<int, 2>{});
  };
  // (body sizes up to what the heads and the launch's k leave room for: G positions)
  auto run = [&](auto n) __attribute__((always_inline)) {
    constexpr int N = decltype(n)::value;
    if constexpr (N <= G) {
      if (span <= N) {
        body(std::integral_constant<int, N>{});
        return true;
      }
    }
    return false;
  };
  if (run(std::integral_constant<int, 4>{}) || run(std::integral_constant<int, 8>{}) || run(std::integral_constant<int, 12>{}) ||
      run(std::integral_constant<int, 16>{}) || run(std::integral_constant<int, 20>{}) || run(std::integral_constant<int, 24>{}))
    return;
  body(std::integral_constant<int, G>{});
}

// ---------------------------------------------------------------------------
// Register-resident backward row (round 5, after the counters: the waves of the kernel above are parked in s_waitcnt for 54-60 % of
// their cycles -- a row is ~12 dependent global round trips: it reads every feature TWICE, 4 bytes per lane and load, and each column
// chunk's loads wait behind the previous chunk's stores).  The forward's layout instead (tgat_attn_reduce_reg_kernel): a lane owns its
// float4 column of the edge features, its neighbor-feature column and its two Time2Vec columns of EVERY covered slot in registers.  A row
// is then THREE round trips -- (1) dzbar, the folded query, slot times / ids, attention weights; (2) every feature of the span, all loads
// in flight together; (3) stores -- with every store after the last load.  The cosines and sines of the lane's own two time columns are
// staged in LDS between the passes (own-lane write, own-lane read: no barrier, no cross-lane traffic); scores, the softmax backward and
// the weights' broadcast travel by DPP / v_readlane.
// Shapes: d <= 64 (one neighbor column per lane: the leaf layers), D % 4 == 0 with D / 4 <= 64, T <= 128, k <= G; others take the kernel
// above.  Sums are ordered like the forward's (column-in-lane, then xor 32 ... 1): not bit-identical to the kernel above, same tolerance.
// ---------------------------------------------------------------------------
// NBV: the neighbor features are a float4 column per lane too (d % 4 == 0, d / 4 <= 64: the layers above the leaves).  Their values and
// dnbr's old values are 8 more registers per slot: the variant runs ONE wave per SIMD (512 registers) and is launched for few rows only
// (the 600-row layer of the headline step: 600 waves on 1 024 SIMDs -- the launch lasts as long as one row, occupancy buys nothing).
template <int H, int G, bool NBV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NBV ? 1 : 2, NBV ? 1 : 2))) void tgat_attn_backward_reg_kernel(const AttnBwdArgs a) {
  static_assert(G * H <= 64, "lane = slot * H + head must fit the wave");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int lane = lane_id();
  const int k = a.k, T = a.T, d = a.d, D = a.D, Cs = a.Cs, D4 = a.D >> 2;
  float* s_cos = lds_all;        // [k][T]: written and read by the lane that owns the column
  float* s_sin = s_cos + k * T;  // the sines of the same arguments
  const long long r = blockIdx.x;
  if (r >= a.R) return;
  const float* lv_nbrf = a.nbrf;
  const float* lv_ex = a.ex;
  const int32_t* lv_eid = a.eid;
  const float* lv_table = a.table;
  const int64_t *lv_seed_t = a.seed_t, *lv_nbr_t = a.nbr_t;
  float* lv_dnbr = a.dnbr;
  if (a.n_seg > 0) {
    int sg = 0;
    for (int i2 = 1; i2 < a.n_seg; ++i2)
      if (r >= a.seg_begin[i2]) sg = i2;
    lv_nbrf = a.seg_nbrf[sg]; lv_ex = a.seg_ex[sg]; lv_eid = a.seg_eid[sg]; lv_table = a.seg_table[sg];
    lv_seed_t = a.seg_seed_t[sg]; lv_nbr_t = a.seg_nbr_t[sg]; lv_dnbr = a.seg_dnbr[sg];
  }
  const float* __restrict__ q = a.qf + r * (long long)H * Cs;
  const float* __restrict__ dz = a.dzbar + r * (long long)H * Cs;
  float* __restrict__ dq = a.dqf + r * (long long)H * Cs;
  const float* __restrict__ nb = lv_nbrf + r * (long long)k * d;
  const float4* __restrict__ ex4 = lv_ex ? reinterpret_cast<const float4*>(lv_ex + r * (long long)k * D) : nullptr;
  const float4* __restrict__ table4 = reinterpret_cast<const float4*>(lv_table);
  float* dnb = lv_dnbr ? lv_dnbr + r * (long long)k * d : nullptr;
  const int d4 = d >> 2;
  const bool e_on = lane < D4, n_on = NBV ? lane < d4 : lane < d, t0_on = lane < T, t1_on = lane + kWave < T;

  // ---- round trip 1: everything that depends on the row number only, issued together (dzbar first: the early exit waits for it alone) ----
  float4 ge[H], qe[H], gn[H], qn[H];  // (gn, qn: .x only unless NBV)
  float gt0[H], gt1[H], qt0[H], qt1[H];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float* gh = dz + h * Cs;
    ge[h] = e_on ? make_float4(gh[d + 4 * lane], gh[d + 4 * lane + 1], gh[d + 4 * lane + 2], gh[d + 4 * lane + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (NBV) gn[h] = n_on ? make_float4(gh[4 * lane], gh[4 * lane + 1], gh[4 * lane + 2], gh[4 * lane + 3]) : zero4;
    else gn[h] = make_float4(n_on ? gh[lane] : 0.f, 0.f, 0.f, 0.f);
    gt0[h] = t0_on ? gh[d + D + lane] : 0.f;
    gt1[h] = t1_on ? gh[d + D + lane + kWave] : 0.f;
  }
  float my_dt = 0.f;
  int my_eid = -1;
  if (lane < k) {
    my_dt = (float)(lv_seed_t[r] - lv_nbr_t[r * k + lane]);  // int64 subtract, then round-to-nearest f32 (tgat.py:143-145)
    if (lv_eid) my_eid = lv_eid[r * k + lane];
  }
  const int js = lane / H, jh = lane - js * H;  // lane j = slot * H + head
  const bool live = js < k;
  const float A = live ? a.probs[r * (long long)H * k + jh * k + js] : 0.f;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float* qh = q + h * Cs;
    qe[h] = e_on ? make_float4(qh[d + 4 * lane], qh[d + 4 * lane + 1], qh[d + 4 * lane + 2], qh[d + 4 * lane + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (NBV) qn[h] = n_on ? make_float4(qh[4 * lane], qh[4 * lane + 1], qh[4 * lane + 2], qh[4 * lane + 3]) : zero4;
    else qn[h] = make_float4(n_on ? qh[lane] : 0.f, 0.f, 0.f, 0.f);
    qt0[h] = t0_on ? qh[d + D + lane] : 0.f;
    qt1[h] = t1_on ? qh[d + D + lane + kWave] : 0.f;
  }
  const float w0 = t0_on ? a.tw[lane] : 0.f, b0 = t0_on ? a.tb[lane] : 0.f;
  const float w1 = t1_on ? a.tw[lane + kWave] : 0.f, b1 = t1_on ? a.tb[lane + kWave] : 0.f;

  // ---- rows without upstream gradient (see the kernel above): zeros out, no feature read ----
  {
    bool nz = false;
#pragma unroll
    for (int h = 0; h < H; ++h)
      nz = nz || ge[h].x != 0.f || ge[h].y != 0.f || ge[h].z != 0.f || ge[h].w != 0.f || gn[h].x != 0.f || gn[h].y != 0.f || gn[h].z != 0.f || gn[h].w != 0.f || gt0[h] != 0.f || gt1[h] != 0.f;
    if (!__any(nz)) {
      for (int c = lane; c < H * Cs; c += kWave) dq[c] = 0.f;
      for (int c = lane; c < 2 * T; c += kWave) a.dtime[r * 2LL * T + c] = 0.f;
      return;
    }
  }
  // the slots that carry gradient: from the first one with A != 0 in some head to the end (see the kernel above)
  const unsigned long long amask = __ballot(A != 0.f);
  const int span = amask ? k - (__ffsll((long long)amask) - 1) / H : k;
  // the cosine's reduction path, once per row: a sufficient test first, the exact one only if it fails (tgat.hip: row_args_small)
  bool row_small;
  {
    const float m = lanes::wave_max(fabsf(my_dt));
    const float lim = 0.99f * kCosSmallLimit;
    row_small = __all(__fmaf_rn(m, fabsf(w0), fabsf(b0)) < lim && __fmaf_rn(m, fabsf(w1), fabsf(b1)) < lim);
    if (!row_small) {
      bool small = true;
      for (int s = 0; s < k; ++s) {
        const float dt = lanes::bcast(my_dt, s);
        small = small && fabsf(__fmaf_rn(dt, w0, b0)) < kCosSmallLimit && fabsf(__fmaf_rn(dt, w1, b1)) < kCosSmallLimit;
      }
      row_small = __all(small);
    }
  }
  const float mk = live ? dropout_scale(a.drop, (unsigned long long)(a.drop_row0 + r) * (H * k) + jh * k + js) : 0.f;

  auto body = [&](auto gsc, auto smallc) __attribute__((always_inline)) {
    constexpr int GS = decltype(gsc)::value;  // slots [k - GS, k) (k < GS: slots [0, k), the positions past k re-read slot k - 1 with weight 0)
    constexpr bool SMALL = decltype(smallc)::value;
    constexpr int NV = GS * H, NVp = NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;
    const int s0 = k > GS ? k - GS : 0;
    // ---- round trip 2: the span's features, all in flight together; dnbr's old values with them (it is accumulated into) ----
    float4 ze[GS], zs[GS], dn[GS];  // (zs, dn: .x only unless NBV)
#pragma unroll
    for (int i = 0; i < GS; ++i) {
      const int sl = s0 + i < k ? s0 + i : k - 1;
      if (lv_eid) {  // wave-uniform: the slot's row of the resident store, zeros for a pad slot
        const int e = lanes::bcast(my_eid, sl);
        ze[i] = (e_on && e >= 0) ? table4[(long long)e * D4 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        ze[i] = e_on ? ex4[sl * D4 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (NBV) {
        zs[i] = n_on ? reinterpret_cast<const float4*>(nb + (long long)sl * d)[lane] : zero4;
        dn[i] = (dnb && n_on) ? reinterpret_cast<const float4*>(dnb + (long long)sl * d)[lane] : zero4;
      } else {
        zs[i] = make_float4(n_on ? nb[(long long)sl * d + lane] : 0.f, 0.f, 0.f, 0.f);
        dn[i] = make_float4((dnb && n_on) ? dnb[(long long)sl * d + lane] : 0.f, 0.f, 0.f, 0.f);
      }
    }
    // ---- pass 1: dA[h][s] = dzbar[h] . z[s]; the lane's cosines and sines go to LDS on the way ----
    float P[NVp];
#pragma unroll
    for (int j = 0; j < NVp; ++j) P[j] = 0.f;
#pragma unroll
    for (int i = 0; i < GS; ++i) {
      const int sl = s0 + i < k ? s0 + i : k - 1;
      const float dt = lanes::bcast(my_dt, sl);
      float sn0, c0, sn1, c1;
      lanes::sincos_path<SMALL>(__fmaf_rn(dt, w0, b0), sn0, c0);
      lanes::sincos_path<SMALL>(__fmaf_rn(dt, w1, b1), sn1, c1);
      if (t0_on) { s_cos[sl * T + lane] = c0; s_sin[sl * T + lane] = sn0; }
      if (t1_on) { s_cos[sl * T + lane + kWave] = c1; s_sin[sl * T + lane + kWave] = sn1; }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float p = ge[h].x * ze[i].x;
        p = __fmaf_rn(ge[h].y, ze[i].y, p);
        p = __fmaf_rn(ge[h].z, ze[i].z, p);
        p = __fmaf_rn(ge[h].w, ze[i].w, p);
        p = __fmaf_rn(gn[h].x, zs[i].x, p);
        if (NBV) {
          p = __fmaf_rn(gn[h].y, zs[i].y, p);
          p = __fmaf_rn(gn[h].z, zs[i].z, p);
          p = __fmaf_rn(gn[h].w, zs[i].w, p);
        }
        p = __fmaf_rn(gt0[h], c0, p);  // (a lane past T: its dzbar column is 0)
        p = __fmaf_rn(gt1[h], c1, p);
        P[i * H + h] = p;
      }
    }
    float sc = lanes::reduce_scatter<NVp>(P, lane);  // lane L: entry L mod NVp = (position i, head h), i * H + h
    if (s0 > 0) sc = __shfl(sc, (lane - s0 * H) & (NVp - 1));  // ... to the lane of the ORIGINAL slot
    // ---- softmax backward in the lanes (slot, head); zbar used A' = A * mk (dropout on the softmax output, attention.py:119) ----
    const float dA = (live && js >= s0) ? sc * mk : 0.f;
    const float dot = lanes::butterfly_sum<H>(A * dA);
    const float ds = A * (dA - dot);  // a masked slot has A == 0
    const float Ap = A * mk;

    // ---- pass 2, from registers: dqf[h] = scale * sum_s ds[h][s] z[s];  dz[s] = sum_h A' dzbar[h] + scale * ds * qf[h] ----
    float4 ae[H], an[H];
    float at0[H], at1[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      ae[h] = an[h] = zero4;
      at0[h] = at1[h] = 0.f;
    }
    float dw0 = 0.f, db0 = 0.f, dw1 = 0.f, db1 = 0.f;
#pragma unroll
    for (int i = 0; i < GS; ++i) {
      const int sp = s0 + i;                 // (sp >= k: a position past the row's slots -- its lanes are not live, every weight is 0)
      const int sl = sp < k ? sp : k - 1;
      const float c0 = t0_on ? s_cos[sl * T + lane] : 0.f, sn0 = t0_on ? s_sin[sl * T + lane] : 0.f;
      const float c1 = t1_on ? s_cos[sl * T + lane + kWave] : 0.f, sn1 = t1_on ? s_sin[sl * T + lane + kWave] : 0.f;
      const float dt = lanes::bcast(my_dt, sl);
      float4 dzn = zero4;
      float dzt0 = 0.f, dzt1 = 0.f;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float wd = lanes::bcast(ds, sp * H + h), wa = lanes::bcast(Ap, sp * H + h);
        ae[h].x = __fmaf_rn(wd, ze[i].x, ae[h].x); ae[h].y = __fmaf_rn(wd, ze[i].y, ae[h].y);
        ae[h].z = __fmaf_rn(wd, ze[i].z, ae[h].z); ae[h].w = __fmaf_rn(wd, ze[i].w, ae[h].w);
        an[h].x = __fmaf_rn(wd, zs[i].x, an[h].x);
        if (NBV) { an[h].y = __fmaf_rn(wd, zs[i].y, an[h].y); an[h].z = __fmaf_rn(wd, zs[i].z, an[h].z); an[h].w = __fmaf_rn(wd, zs[i].w, an[h].w); }
        at0[h] = __fmaf_rn(wd, c0, at0[h]);
        at1[h] = __fmaf_rn(wd, c1, at1[h]);
        const float sq = a.scale * wd;
        dzn.x += wa * gn[h].x + sq * qn[h].x;
        if (NBV) { dzn.y += wa * gn[h].y + sq * qn[h].y; dzn.z += wa * gn[h].z + sq * qn[h].z; dzn.w += wa * gn[h].w + sq * qn[h].w; }
        dzt0 += wa * gt0[h] + sq * qt0[h];
        dzt1 += wa * gt1[h] + sq * qt1[h];
      }
      dn[i].x += dzn.x;
      if (NBV) { dn[i].y += dzn.y; dn[i].z += dzn.z; dn[i].w += dzn.w; }
      const float g0 = -sn0 * dzt0, g1 = -sn1 * dzt1;  // d cos(arg) / d arg
      dw0 = __fmaf_rn(g0, dt, dw0); db0 += g0;
      dw1 = __fmaf_rn(g1, dt, dw1); db1 += g1;
    }
    // ---- round trip 3: the stores, after the last load ----
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float* zh = dq + h * Cs;
      if (NBV) {
        if (n_on) { zh[4 * lane] = an[h].x * a.scale; zh[4 * lane + 1] = an[h].y * a.scale; zh[4 * lane + 2] = an[h].z * a.scale; zh[4 * lane + 3] = an[h].w * a.scale; }
      } else if (n_on) {
        zh[lane] = an[h].x * a.scale;
      }
      if (e_on) {
        zh[d + 4 * lane] = ae[h].x * a.scale; zh[d + 4 * lane + 1] = ae[h].y * a.scale;
        zh[d + 4 * lane + 2] = ae[h].z * a.scale; zh[d + 4 * lane + 3] = ae[h].w * a.scale;
      }
      if (t0_on) zh[d + D + lane] = at0[h] * a.scale;
      if (t1_on) zh[d + D + lane + kWave] = at1[h] * a.scale;
    }
    if (dnb && n_on) {
#pragma unroll
      for (int i = 0; i < GS; ++i)
        if (s0 + i < k) {
          if (NBV) reinterpret_cast<float4*>(dnb + (long long)(s0 + i) * d)[lane] = dn[i];
          else dnb[(long long)(s0 + i) * d + lane] = dn[i].x;
        }
    }
    float* dtr = a.dtime + r * 2LL * T;
    if (t0_on) { dtr[lane] = dw0; dtr[T + lane] = db0; }
    if (t1_on) { dtr[lane + kWave] = dw1; dtr[T + lane + kWave] = db1; }
  };
  auto run = [&](auto n) __attribute__((always_inline)) {
    if (row_small) body(n, std::true_type{});
    else body(n, std::false_type{});
  };
  if constexpr (G > 16) {
    if (span <= 4) return run(std::integral_constant<int, 4>{});
    if (span <= 8) return run(std::integral_constant<int, 8>{});
    if (span <= 12) return run(std::integral_constant<int, 12>{});
    if (span <= 16) return run(std::integral_constant<int, 16>{});
  } else if constexpr (G > 8) {
    if (span <= 4) return run(std::integral_constant<int, 4>{});
    if (span <= 8) return run(std::integral_constant<int, 8>{});
  }
  run(std::integral_constant<int, G>{});
}

}  // namespace tgmx

using namespace tgmx;

static int pick_splits(long long R, long long tiles) {
  // enough (tile, split) work items to fill ~1024 SIMDs, at least 64 rows per split
  long long s = (2048 + tiles - 1) / tiles;
  const long long max_s = (R + 63) / 64;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return (int)s;
}

extern "C" size_t tgmx_sgemm_tn_workspace_bytes(int64_t R, int32_t M, int32_t N, int32_t batch) {
  const long long tiles = (long long)((M + 31) / 32) * ((N + 63) / 64) * batch;
  return (size_t)pick_splits(R, tiles) * (size_t)M * N * batch * sizeof(float);
}

extern "C" int tgmx_sgemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t R, int32_t M,
                             int32_t N, int32_t batch, int64_t strideA, int64_t strideB, int64_t strideC, int32_t accumulate,
                             float* workspace, tgmx_stream_t stream) {
  TGMX_REQUIRE(R >= 0 && M > 0 && N > 0 && batch > 0 && lda >= M && ldb >= N && ldc >= N, "sgemm_tn: bad sizes");
  TGMX_REQUIRE(A && B && C && workspace, "sgemm_tn: null pointer");
  const long long tiles = (long long)((M + 31) / 32) * ((N + 63) / 64) * batch;
  const int splits = R > 0 ? pick_splits(R, tiles) : 1;
  GemmTnArgs g{A, B, workspace, lda, ldb, strideA, strideB, R, R > 0 ? (R + splits - 1) / splits : 0, M, N, splits};
  g.rows_per_split = (g.rows_per_split + 7) / 8 * 8;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sgemm_tn_kernel, dim3((unsigned)((M + 127) / 128), (unsigned)((N + 63) / 64), (unsigned)(batch * splits)), dim3(256), 0, st, g);
  const long long MN = (long long)M * N;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((MN + 255) / 256), (unsigned)batch), dim3(256), 0, st, workspace, splits, M, N, C,
                     (long long)ldc, (long long)strideC, accumulate);
  TGMX_CHECK_LAUNCH("sgemm_tn");
  return TGMX_OK;
}

extern "C" int tgmx_colsum(const float* in, int64_t ld, int64_t R, int32_t C, float* out, int32_t accumulate, float* workspace,
                           tgmx_stream_t stream) {
  TGMX_REQUIRE(R >= 0 && C > 0 && ld >= C, "colsum: bad sizes");
  TGMX_REQUIRE(in && out && workspace, "colsum: null pointer");
  int splits = (int)((R + 63) / 64);
  if (splits > 256) splits = 256;
  if (splits < 1) splits = 1;
  const long long rps = (R + splits - 1) / splits;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)splits), dim3(256), 0, st, in, (long long)ld,
                     (long long)R, C, rps, workspace);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((C + 255) / 256), 1), dim3(256), 0, st, workspace, splits, 1, C, out, (long long)C,
                     0LL, accumulate);
  TGMX_CHECK_LAUNCH("colsum");
  return TGMX_OK;
}

extern "C" int tgmx_relu_mask(float* grad, int64_t ldg, const float* act, int64_t lda, int64_t R, int32_t C, tgmx_stream_t stream) {
  TGMX_REQUIRE(R >= 0 && C > 0, "relu_mask: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(grad && act, "relu_mask: null pointer");
  long long blocks = (R * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, (long long)ldg, act, (long long)lda,
                     (long long)R, C);
  TGMX_CHECK_LAUNCH("relu_mask");
  return TGMX_OK;
}

extern "C" int tgmx_add_cols(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t R, int32_t C, int32_t accumulate,
                             tgmx_stream_t stream) {
  TGMX_REQUIRE(R >= 0 && C > 0, "add_cols: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(dst && src, "add_cols: null pointer");
  long long blocks = (R * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(add_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst, (long long)ldd, src, (long long)lds,
                     (long long)R, C, accumulate);
  TGMX_CHECK_LAUNCH("add_cols");
  return TGMX_OK;
}

extern "C" int tgmx_ln_backward(const float* dout, int64_t ldd, const float* y, int64_t ldy, const float* res, int64_t ldr,
                                const float* gamma, int32_t O, float eps, int64_t R, float* du, int64_t ldu, float* dgx, int64_t ldg,
                                tgmx_stream_t stream) {
  TGMX_REQUIRE(O > 0 && R >= 0, "ln_backward: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(dout && y && res && gamma && du && dgx, "ln_backward: null pointer");
  hipLaunchKernelGGL(ln_backward_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dout, (long long)ldd, y,
                     (long long)ldy, res, (long long)ldr, gamma, O, eps, (long long)R, du, (long long)ldu, dgx, (long long)ldg);
  TGMX_CHECK_LAUNCH("ln_backward");
  return TGMX_OK;
}

static int attn_backward_impl(const float* qf, const float* probs, const float* dzbar, const float* nbrf, int32_t d, const float* ex,
                              const int32_t* eid, const float* table, int32_t D, const int64_t* seed_t, const int64_t* nbr_t, const float* tw,
                              const float* tb, int32_t T, int32_t H, int32_t k, int64_t R, float scale, int32_t head_stride, float* dqf,
                              float* dnbr, float* dtime_rows, const tgmx_dropout_t* drop, tgmx_stream_t stream);
static int launch_attn_backward(const AttnBwdArgs& a, int H, tgmx_stream_t stream);

extern "C" int tgmx_tgat_attn_backward(const float* qf, const float* probs, const float* dzbar, const float* nbrf, int32_t d,
                                       const float* ex, int32_t D, const int64_t* seed_t, const int64_t* nbr_t, const float* tw,
                                       const float* tb, int32_t T, int32_t H, int32_t k, int64_t R, float scale, int32_t head_stride,
                                       float* dqf, float* dnbr, float* dtime_rows, const tgmx_dropout_t* drop, tgmx_stream_t stream) {
  return attn_backward_impl(qf, probs, dzbar, nbrf, d, ex, nullptr, nullptr, D, seed_t, nbr_t, tw, tb, T, H, k, R, scale, head_stride, dqf, dnbr,
                            dtime_rows, drop, stream);
}

static int attn_backward_impl(const float* qf, const float* probs, const float* dzbar, const float* nbrf, int32_t d, const float* ex,
                              const int32_t* eid, const float* table, int32_t D, const int64_t* seed_t, const int64_t* nbr_t, const float* tw,
                              const float* tb, int32_t T, int32_t H, int32_t k, int64_t R, float scale, int32_t head_stride, float* dqf,
                              float* dnbr, float* dtime_rows, const tgmx_dropout_t* drop, tgmx_stream_t stream) {
  TGMX_REQUIRE(d > 0 && D >= 0 && T > 0 && k > 0 && R >= 0, "tgat_attn_backward: bad sizes");
  TGMX_REQUIRE((H == 1 || H == 2 || H == 4 || H == 8) && k * H <= 64, "tgat_attn_backward: needs n_heads in {1,2,4,8} and k * n_heads <= 64 (k=%d, H=%d)", k, H);
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(qf && probs && dzbar && nbrf && (D == 0 || ex || (eid && table)) && seed_t && nbr_t && tw && tb && dqf && dtime_rows,
               "tgat_attn_backward: null pointer");
  const int C = d + D + T;
  AttnBwdArgs a{qf, probs, dzbar, nbrf, ex, ex ? nullptr : eid, ex ? nullptr : table, seed_t, nbr_t, tw, tb, dqf, dnbr, dtime_rows, R, d, D, T, k, C,
                head_stride ? head_stride : C, scale};
  a.drop = make_dropout(drop);
  a.drop_row0 = drop ? drop->row0 : 0;
  return launch_attn_backward(a, H, stream);
}

static int launch_attn_backward(const AttnBwdArgs& a, int H, tgmx_stream_t stream) {
  const int k = a.k, T = a.T;
  const long long R = a.R;
  const size_t per_wave = (2 * (size_t)k * T + 2 * (size_t)k + 2 * (size_t)H * k) * sizeof(float);
  static const int waves_knob = [] { const char* e = getenv("TGMX_ATTN_BWD_WPB"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 1; }();
  // one wave per workgroup: the waves share nothing, and a multi-wave workgroup's wave slots only refill when its SLOWEST row is done
  // (rows differ 10 x in cost: zero-upstream rows end at once); TGMX_ATTN_BWD_WPB = 2 / 4 is the A/B knob
  int waves = waves_knob;
  while (waves > 1 && per_wave * waves > 64 * 1024) waves >>= 1;
  TGMX_REQUIRE(per_wave * waves <= 64 * 1024, "tgat_attn_backward: k*T=%d too large for LDS", k * T);
  const dim3 grid((unsigned)((R + waves - 1) / waves)), block(waves * kWave);
  hipStream_t st = (hipStream_t)stream;
  // the register-resident row (tgat_attn_backward_reg_kernel) for the shapes it holds; TGMX_ATTN_BWD_REG=0: the A/B knob
  static const bool reg_knob = [] { const char* e = getenv("TGMX_ATTN_BWD_REG"); return !(e && e[0] == '0'); }();
  // (the float4-neighbor variant runs one wave per SIMD: for launches that do not fill the SIMDs twice anyway)
  static const bool nbv_knob = [] { const char* e = getenv("TGMX_ATTN_BWD_REG_NBV"); return !(e && e[0] == '0'); }();
  const bool nbv = nbv_knob && a.d > 64 && a.d % 4 == 0 && a.d / 4 <= 64 && R <= 2048;
  if (reg_knob && (H == 1 || H == 2) && k <= 20 && (a.d <= 64 || nbv) && a.D > 0 && a.D % 4 == 0 && a.D / 4 <= 64 && T <= 128) {
    uintptr_t bits = (uintptr_t)a.ex | (uintptr_t)a.table;  // D % 4 == 0: a biased level pointer keeps its alignment
    if (nbv) bits |= (uintptr_t)a.nbrf | (uintptr_t)a.dnbr;
    for (int i = 0; i < a.n_seg; ++i) {
      bits |= (uintptr_t)a.seg_ex[i] | (uintptr_t)a.seg_table[i];
      if (nbv) bits |= (uintptr_t)a.seg_nbrf[i] | (uintptr_t)a.seg_dnbr[i];
    }
    if ((bits & 15) == 0) {
      const size_t lds = 2 * (size_t)k * T * sizeof(float);
      const dim3 rgrid((unsigned)R), rblock(kWave);
#define TGMX_ATTN_BWD_REG(H_, G_)                                                                             \
  do {                                                                                                       \
    if (nbv) hipLaunchKernelGGL((tgat_attn_backward_reg_kernel<H_, G_, true>), rgrid, rblock, lds, st, a);    \
    else hipLaunchKernelGGL((tgat_attn_backward_reg_kernel<H_, G_, false>), rgrid, rblock, lds, st, a);       \
  } while (0)
      if (H == 1 && k <= 10) TGMX_ATTN_BWD_REG(1, 10);
      else if (H == 1) TGMX_ATTN_BWD_REG(1, 20);
      else if (k <= 10) TGMX_ATTN_BWD_REG(2, 10);
      else TGMX_ATTN_BWD_REG(2, 20);
#undef TGMX_ATTN_BWD_REG
      TGMX_CHECK_LAUNCH("tgat_attn_backward");
      return TGMX_OK;
    }
  }
#define TGMX_ATTN_BWD(H_)                                                                                        \
  do {                                                                                                          \
    if (k <= 20) hipLaunchKernelGGL((tgat_attn_backward_kernel<H_, 20>), grid, block, per_wave * waves, st, a); \
    else hipLaunchKernelGGL((tgat_attn_backward_kernel<H_, 64>), grid, block, per_wave * waves, st, a);          \
  } while (0)
  switch (H) {
    case 1: TGMX_ATTN_BWD(1); break;
    case 2: TGMX_ATTN_BWD(2); break;
    case 4: hipLaunchKernelGGL((tgat_attn_backward_kernel<4, 64>), grid, block, per_wave * waves, st, a); break;
    default: hipLaunchKernelGGL((tgat_attn_backward_kernel<8, 64>), grid, block, per_wave * waves, st, a); break;
  }
#undef TGMX_ATTN_BWD
  TGMX_CHECK_LAUNCH("tgat_attn_backward");
  return TGMX_OK;
}

// ---------------------------------------------------------------------------
// The whole backward as one call: tgm_amd/nn/_tgat_train.py's composition, in its order, from C++.
// ---------------------------------------------------------------------------
namespace tgmx {
// d_tb[c] -= sin(tb[c]) * g[c]  (the residual's time columns are cos(tb): rres = [x | 0 | cos(tb)]); two roundings, like the
// torch expression it replaces (no contraction into an fma)
__global__ __launch_bounds__(256) void time_bias_residual_kernel(float* __restrict__ d_tb, const float* __restrict__ tb,
                                                                 const float* __restrict__ g, int T) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < T) d_tb[c] = __fsub_rn(d_tb[c], __fmul_rn(sinf(tb[c]), g[c]));
}

// ---- every split reduction of a layer's backward in two launches ----
// The composed path follows each weight-gradient GEMM with its own reduce_partials launch and runs two launches per bias /
// LayerNorm / Time2Vec column sum: 25 launches per layer, 5-9 us each on a handful of CUs.  Here the GEMMs only leave their split
// partials behind; at the end of the layer ONE launch forms all the column sums' partials and ONE reduces everything, each job
// with the arithmetic (split counts, summation order) of the kernels above: bit-identical results.
constexpr int kBatchJobs = 48;
struct ReduceJob {
  const float* partial;
  float* out;
  long long ldo;
  int splits, M, N, accumulate;
};
struct ReduceJobs {
  ReduceJob job[kBatchJobs];
};
__global__ __launch_bounds__(256) void reduce_batch_kernel(const ReduceJobs J) {
  const ReduceJob j = J.job[blockIdx.y];
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long MN = (long long)j.M * j.N;
  if (e >= MN) return;
  const float* p = j.partial + e;
  float acc = 0.f;
  for (int s = 0; s < j.splits; s += 8) {  // as reduce_partials_kernel
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s + u < j.splits ? p[(long long)(s + u) * MN] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  const int m = (int)(e / j.N), n = (int)(e - (long long)m * j.N);
  float* o = j.out + (long long)m * j.ldo + n;
  *o = j.accumulate ? *o + acc : acc;
}
struct ColsumJob {
  const float* in;
  float* partial;
  long long ld, R, rows_per_split;
  int C, splits;
};
struct ColsumJobs {
  ColsumJob job[kBatchJobs];
};
__global__ __launch_bounds__(256) void colsum_batch_kernel(const ColsumJobs J) {
  const ColsumJob j = J.job[blockIdx.z];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= j.C || (int)blockIdx.y >= j.splits) return;
  const long long r0 = (long long)blockIdx.y * j.rows_per_split;
  long long r1 = r0 + j.rows_per_split;
  if (r1 > j.R) r1 = j.R;
  float acc = 0.f;
  for (long long r = r0; r < r1; r += 8) {  // as colsum_partial_kernel
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = r + u < r1 ? j.in[(r + u) * j.ld + c] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  j.partial[(long long)blockIdx.y * j.C + c] = acc;
}
}  // namespace tgmx

static int pick_splits(long long R, long long tiles);

namespace {
struct Bump {
  float* base;
  size_t off;
  float* take(size_t n) {
    const size_t p = off;
    off += (n + 63) & ~(size_t)63;  // 256-byte granules
    return base + p;
  }
};

// One pass over the backward; dry: only the workspace layout is computed (floats needed -> *need_floats)
int tgat_backward_pass(const tgmx_tgat_model_t* m, const tgmx_tgat_layout_t* lay, const tgmx_tgat_hop_t* hops, const float* saved,
                       const float* dz, long long ldz, const tgmx_dropout_t* drop, const tgmx_tgat_grads_t* g, float* ws,
                       hipStream_t st, bool dry, size_t* need_floats) {
  const int L = m->num_layers, d0 = m->d0, T = m->layers[0].T;
  const float p_drop = drop ? drop->p : 0.f;
  static const tgmx_tgat_grads_t no_grads{};
  if (!g) g = &no_grads;  // (dry pass: nothing below is dereferenced)
  tgmx_stream_t stream = (tgmx_stream_t)st;
  Bump b{ws, 0};
  static const bool sync_each = getenv("TGMX_BWD_SYNC") != nullptr;  // diagnosis: name every step and wait for it
  // (tn sizes its scratch in the dry pass: evaluated in both)
#define RUN_ALWAYS(call)                                      \
  do {                                                        \
    if (!dry && sync_each) fprintf(stderr, "[bwd] %s\n", #call); \
    const int rc_ = (call);                                   \
    if (rc_) return rc_;                                      \
    if (!dry && sync_each) (void)hipStreamSynchronize(st);    \
  } while (0)
#define RUN(call)                                             \
  do {                                                        \
    if (!dry) {                                               \
      if (sync_each) fprintf(stderr, "[bwd] %s\n", #call);    \
      const int rc_ = (call);                                 \
      if (rc_) return rc_;                                    \
      if (sync_each) (void)hipStreamSynchronize(st);          \
    }                                                         \
  } while (0)
  static const bool batched = !(getenv("TGMX_BWD_BATCH") && atoi(getenv("TGMX_BWD_BATCH")) == 0);  // A/B knob (0: a reduction launch per job)
  tgmx::ReduceJobs rj;
  tgmx::ColsumJobs cj;
  tgmx::GemmTnJobs tj;
  tj.n = 0;
  tj.first_block[0] = 0;
  int n_rj = 0, n_cj = 0;
  long long rj_most = 0;
  int cj_wide = 0, cj_splits = 0;
  // the column sums' partials, then every reduction queued so far
  auto flush = [&]() -> int {
    if (!dry) {
      if (tj.n) {
        hipLaunchKernelGGL(tgmx::sgemm_tn_batch_kernel, dim3((unsigned)tj.first_block[tj.n]), dim3(256), 0, st, tj);
        TGMX_CHECK_LAUNCH("tgat_backward (weight gradients)");
      }
      if (n_cj) {
        hipLaunchKernelGGL(tgmx::colsum_batch_kernel, dim3((unsigned)((cj_wide + 255) / 256), (unsigned)cj_splits, (unsigned)n_cj), dim3(256), 0, st, cj);
        TGMX_CHECK_LAUNCH("tgat_backward (column sums)");
      }
      if (n_rj) {
        hipLaunchKernelGGL(tgmx::reduce_batch_kernel, dim3((unsigned)((rj_most + 255) / 256), (unsigned)n_rj), dim3(256), 0, st, rj);
        TGMX_CHECK_LAUNCH("tgat_backward (reductions)");
      }
      if (sync_each) (void)hipStreamSynchronize(st);
    }
    n_rj = n_cj = 0;
    tj.n = 0;
    rj_most = 0;
    cj_wide = cj_splits = 0;
    return TGMX_OK;
  };
  auto queue_reduce = [&](const float* partial, int splits, int M, int N, float* out, long long ldo, int accumulate) -> int {
    if (n_rj == tgmx::kBatchJobs) {
      const int rc = flush();
      if (rc) return rc;
    }
    rj.job[n_rj++] = tgmx::ReduceJob{partial, out, ldo, splits, M, N, accumulate};
    rj_most = (long long)M * N > rj_most ? (long long)M * N : rj_most;
    return batched ? TGMX_OK : flush();
  };
  // C[b] = A[b]^T B[b]: the split partials now (tgmx_sgemm_tn's kernel and split count), the reduction with the layer's others
  auto tn = [&](const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, long long R, int M, int N, int batch,
                long long sA, long long sB, long long sC) -> int {
    const long long tiles = (long long)((M + 31) / 32) * ((N + 63) / 64) * batch;
    const int splits = R > 0 ? pick_splits(R, tiles) : 1;
    float* partial = b.take((size_t)splits * M * N * batch);
    if (tj.n == tgmx::kTnJobs || n_rj + batch > tgmx::kBatchJobs) {
      const int rc = flush();
      if (rc) return rc;
    }
    {  // queued: A and B must stay as they are until the layer's flush (they do: see the call sites)
      tgmx::GemmTnArgs g{A, B, partial, lda, ldb, sA, sB, R, R > 0 ? (R + splits - 1) / splits : 0, M, N, splits};
      g.rows_per_split = (g.rows_per_split + 7) / 8 * 8;
      const int gx = (M + 127) / 128, gy = (N + 63) / 64;
      tj.job[tj.n] = g;
      tj.gx[tj.n] = gx;
      tj.gy[tj.n] = gy;
      tj.first_block[tj.n + 1] = tj.first_block[tj.n] + gx * gy * batch * splits;
      ++tj.n;
    }
    for (int bi = 0; bi < batch; ++bi) {
      const int rc = queue_reduce(partial + (long long)bi * splits * M * N, splits, M, N, C + (long long)bi * sC, ldc, 0);
      if (rc) return rc;
    }
    return TGMX_OK;
  };
  auto nt = [&](const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, long long M, int N, int K, int batch,
                long long sA, long long sB, long long sC) -> int {
    if (dry) return TGMX_OK;
    return tgmx_sgemm_nt(A, lda, B, ldb, C, ldc, M, N, K, nullptr, 0, batch, sA, sB, sC, stream);
  };
  // out[c] (+)= sum_r X[r, c]: queued (tgmx_colsum's split count); X must stay as it is until the layer's flush
  auto colsum = [&](const float* X, long long ld, long long R, int C, float* out, int accumulate) -> int {
    int splits = (int)((R + 63) / 64);
    if (splits > 256) splits = 256;
    if (splits < 1) splits = 1;
    float* partial = b.take((size_t)splits * C);
    if (n_cj == tgmx::kBatchJobs) {
      const int rc = flush();
      if (rc) return rc;
    }
    cj.job[n_cj++] = tgmx::ColsumJob{X, partial, ld, R, (R + splits - 1) / splits, C, splits};
    cj_wide = C > cj_wide ? C : cj_wide;
    cj_splits = splits > cj_splits ? splits : cj_splits;
    return queue_reduce(partial, splits, 1, C, out, (long long)C, accumulate);
  };
  // ---- the weights in the layouts the NT GEMMs below read them in: one launch for all layers ----
  struct LayerW {
    float *F2_t, *F1_t, *WO_t, *WV_t, *WK_p, *WQ_t;
  } W[TGMX_TGAT_MAX_LAYERS];
  tgmx_pack_job_t jobs[TGMX_PACK_MAX_JOBS];
  int n_jobs = 0;
  auto job = [&](const float* src, long long src_ld, float* dst, long long dst_ld, int rows, int cols, int dst_rows, int dst_cols) -> int {
    if (n_jobs == TGMX_PACK_MAX_JOBS) {
      RUN(tgmx_pack2d(jobs, n_jobs, stream));
      n_jobs = 0;
    }
    jobs[n_jobs++] = tgmx_pack_job_t{src, dst, src_ld, dst_ld, rows, cols, dst_rows, dst_cols, 1, 0};  // every one is a transpose
    return TGMX_OK;
  };
  for (int l = 0; l < L; ++l) {
    const tgmx_tgat_layer_t& ly = m->layers[l];
    const tgmx_tgat_layer_layout_t& lo = lay->layers[l];
    const int O = ly.O, H = ly.H, dh = O / H, C = ly.d + ly.D + ly.T, emb = ly.emb, emb_out = ly.emb_out;
    const int Op = lo.Op, dhp = lo.dhp, Cp = lo.Cp, Kc = lo.Kc, Ep = lo.Ep;
    LayerW& w = W[l];
    w.F2_t = b.take((size_t)emb * emb_out);        // [emb, emb_out]      = fc2.weight^T
    w.F1_t = b.take((size_t)(O + d0) * emb);       // [O + d0, emb]       = fc1.weight^T
    w.WO_t = b.take((size_t)H * dhp * O);          // [H * dhp, O]        = W_O^T, head h's dh rows at row h * dhp (zero rows between)
    w.WV_t = b.take((size_t)C * H * dhp);          // [C, H * dhp]        = W_V^T, heads dhp apart
    w.WK_p = b.take((size_t)O * Cp);               // [O, Cp]             = W_K, rows padded
    w.WQ_t = b.take((size_t)O * H * dhp);          // [O, H * dhp]        = W_Q^T, heads dhp apart
    int rc = job(ly.fc2_w, Ep, w.F2_t, emb_out, emb, emb_out, emb, emb_out);
    if (!rc) rc = job(ly.fc1_w, Kc, w.F1_t, emb, O + d0, emb, O + d0, emb);
    for (int h = 0; h < H && !rc; ++h) rc = job(ly.W_O + h * dh, Op, w.WO_t + (long long)h * dhp * O, O, dh, O, dhp, O);
    for (int h = 0; h < H && !rc; ++h) {
      rc = job(ly.W_V + (long long)h * dh * Cp, Cp, w.WV_t + h * dhp, (long long)H * dhp, C, dh, C, dhp);
      if (!rc) rc = job(ly.W_K_t + h * dhp, (long long)H * dhp, w.WK_p + (long long)h * dh * Cp, Cp, dh, C, dh, Cp);
      if (!rc) rc = job(ly.W_Q + (long long)h * dh * Op, Op, w.WQ_t + h * dhp, (long long)H * dhp, O, dh, O, dhp);
    }
    if (rc) return rc;
  }
  if (n_jobs) RUN(tgmx_pack2d(jobs, n_jobs, stream));
  const float* S = saved;
  const float* z0 = S + lay->z0;
  const float* dout = dz;
  long long ld_dout = ldz;
  for (int j = L; j >= 1; --j) {
    const tgmx_tgat_layer_t& ly = m->layers[j - 1];
    const tgmx_tgat_layer_layout_t& lo = lay->layers[j - 1];
    const tgmx_tgat_layer_grads_t& gl = g->layers[j - 1];
    const LayerW& w = W[j - 1];
    const long long R = lo.R;
    const int Op = lo.Op, dhp = lo.dhp, Cp = lo.Cp, Kc = lo.Kc, Ep = lo.Ep;
    const int O = ly.O, H = ly.H, dh = O / H, d = ly.d, D = ly.D, C = d + D + T, emb = ly.emb, emb_out = ly.emb_out;
    const int k = hops[j - 1].k, n_lvl = L - j + 1;
    const float *rres = S + lo.rres, *oattn = S + lo.oattn, *y = S + lo.y, *Q = S + lo.Q, *qf = S + lo.qf, *zbar = S + lo.zbar;
    const float *cat = S + lo.cat, *h1 = S + lo.h1, *probs = S + lo.probs;
    const float* prev = j == 1 ? z0 : S + lay->layers[j - 2].out;  // [level_off[n_lvl + 1], d]
    // ---- merge MLP ----
    RUN_ALWAYS(tn(dout, ld_dout, h1, Ep, gl.fc2_w, emb, R, emb_out, emb, 1, 0, 0, 0));
    RUN_ALWAYS(colsum(dout, ld_dout, R, emb_out, gl.fc2_b, 0));
    float* dh1 = b.take((size_t)R * Ep);
    RUN(nt(dout, ld_dout, w.F2_t, emb_out, dh1, Ep, R, emb, emb_out, 1, 0, 0, 0));
    RUN(tgmx_relu_mask(dh1, Ep, h1, Ep, R, emb, stream));
    RUN_ALWAYS(tn(dh1, Ep, cat, Kc, gl.fc1_w, O + d0, R, emb, O + d0, 1, 0, 0, 0));
    RUN_ALWAYS(colsum(dh1, Ep, R, emb, gl.fc1_b, 0));
    float* dcat = b.take((size_t)R * Kc);
    RUN(nt(dh1, Ep, w.F1_t, emb, dcat, Kc, R, O + d0, emb, 1, 0, 0, 0));
    // ---- LayerNorm(y + rres) ----
    float* du = b.take((size_t)R * Op);
    float* dgx = b.take((size_t)R * Op);
    RUN(tgmx_ln_backward(dcat, Kc, y, Op, rres, Op, ly.ln_g, O, ly.ln_eps, R, du, Op, dgx, Op, stream));
    RUN_ALWAYS(colsum(dgx, Op, R, O, gl.ln_g, 0));
    RUN_ALWAYS(colsum(dcat, Kc, R, O, gl.ln_b, 0));
    // ---- W_O (its output went through dropout: the gradient takes the same mask; the residual branch keeps the unmasked du) ----
    float* du_y = du;
    float* du_masked = b.take((size_t)R * Op);  // (taken with or without dropout: the layout does not depend on the call's arguments)
    if (p_drop > 0.f) {
      du_y = du_masked;
      const tgmx_dropout_t site{p_drop, drop->seed, drop->stream * 64 + 2 * (uint64_t)j + 1, 0};
      RUN(tgmx_dropout(du, Op, R, O, &site, du_y, Op, stream));
    }
    RUN_ALWAYS(tn(du_y, Op, oattn, Op, gl.W_O, O, R, O, O, 1, 0, 0, 0));
    RUN_ALWAYS(colsum(du_y, Op, R, O, gl.b_O, 0));
    // (d oattn with its heads dhp apart, like Q: head 1 of the dense [R, O] layout starts at column dh = 86 -- 8-byte aligned -- and the
    // batched GEMM below then takes the scalar-load path, 32 us instead of 18; same dot products, same order: identical values)
    const long long ld_do = (long long)H * dhp;
    float* doattn = b.take((size_t)R * ld_do);
    RUN(nt(du_y, Op, w.WO_t, O, doattn, ld_do, R, H * dhp, O, 1, 0, 0, 0));
    // ---- W_V fold: oattn[:, head h] = zbar[:, h, :] @ W_V[head h]^T ----
    float* g_WK = gl.W_KV;
    float* g_WV = gl.W_KV + (long long)O * C;
    RUN_ALWAYS(tn(doattn, ld_do, zbar, (long long)H * Cp, g_WV, C, R, dh, C, H, dhp, Cp, (long long)dh * C));
    float* dzbar = b.take((size_t)R * H * Cp);
    RUN(nt(doattn, ld_do, w.WV_t, (long long)H * dhp, dzbar, (long long)H * Cp, R, C, dh, H, dhp, dhp, Cp));
    // ---- per-row attention backward, level by level ----
    float* dqf = b.take((size_t)R * H * Cp);
    float* dtime = b.take((size_t)R * 2 * T);
    const bool need_dprev = j > 1;
    const size_t prev_rows = (size_t)lay->level_off[n_lvl + 1];
    float* dprev = need_dprev ? b.take(prev_rows * d) : nullptr;
    if (need_dprev && !dry) (void)hipMemsetAsync(dprev, 0, prev_rows * d * sizeof(float), st);
    const float scale = (float)pow((double)dh, -0.5);
    {  // every level this layer aggregates in ONE launch (the 600-row levels alone take as long as one row's chain of round trips: 45 us)
      AttnBwdArgs ab{};
      ab.qf = qf; ab.probs = probs; ab.dzbar = dzbar; ab.tw = m->tw; ab.tb = m->tb; ab.dqf = dqf; ab.dtime = dtime;
      ab.R = lay->level_off[n_lvl]; ab.d = d; ab.D = D; ab.T = T; ab.k = k; ab.C = C; ab.Cs = Cp; ab.scale = scale;
      ab.n_seg = n_lvl;
      bool ok = (H == 1 || H == 2 || H == 4 || H == 8) && k * H <= 64;
      for (int i = 0; i < n_lvl; ++i) {
        const long long o = lay->level_off[i], o1 = lay->level_off[i + 1];
        const bool by_id = !hops[i].edge_x && hops[i].nbr_eid && hops[i].edge_table;
        ok = ok && (lay->level_rows[i] == 0 || (hops[i].seed_t && hops[i].nbr_t && (D == 0 || hops[i].edge_x || by_id)));
        ab.seg_begin[i] = o;  // the kernel indexes with the layer-wide row: bias every level's arrays by its first row
        ab.seg_nbrf[i] = prev + o1 * d - o * (long long)k * d;
        ab.seg_ex[i] = hops[i].edge_x ? hops[i].edge_x - o * (long long)k * D : nullptr;
        ab.seg_eid[i] = by_id ? hops[i].nbr_eid - o * (long long)k : nullptr;
        ab.seg_table[i] = by_id ? hops[i].edge_table : nullptr;
        ab.seg_seed_t[i] = hops[i].seed_t - o;
        ab.seg_nbr_t[i] = hops[i].nbr_t - o * (long long)k;
        ab.seg_dnbr[i] = need_dprev ? dprev + o1 * d - o * (long long)k * d : nullptr;
      }
      if (!dry) {
        TGMX_REQUIRE(ok, "tgat_backward: layer %d: unsupported head count / k, or a hop without its sampler tensors", j);
        const tgmx_dropout_t site{p_drop, drop ? drop->seed : 0, (drop ? drop->stream : 0) * 64 + 2 * (uint64_t)j, 0};
        ab.drop = make_dropout(p_drop > 0.f ? &site : nullptr);
        ab.drop_row0 = 0;
      }
      if (ab.R > 0) RUN(launch_attn_backward(ab, H, stream));
    }
    // (the top layer's sums open the Time2Vec gradients -- 0 + x is x, what accumulating into zeroed buffers gave -- the others add to them)
    RUN_ALWAYS(colsum(dtime, 2 * T, R, T, g->tw, j == L ? 0 : 1));
    RUN_ALWAYS(colsum(dtime + T, 2 * T, R, T, g->tb, j == L ? 0 : 1));
    // ---- W_K fold: qf[:, h, :] = Q[:, head h] @ W_K[head h] ----
    RUN_ALWAYS(tn(Q, (long long)H * dhp, dqf, (long long)H * Cp, g_WK, C, R, dh, C, H, dhp, Cp, (long long)dh * C));
    float* dQ = b.take((size_t)R * H * dhp);
    if (!dry) (void)hipMemsetAsync(dQ, 0, (size_t)R * H * dhp * sizeof(float), st);
    RUN(nt(dqf, (long long)H * Cp, w.WK_p, Cp, dQ, (long long)H * dhp, R, dh, C, H, Cp, (long long)dh * Cp, dhp));
    // ---- W_Q: Q[:, head h] = rres @ W_Q[head h rows]^T ----
    RUN_ALWAYS(tn(dQ, (long long)H * dhp, rres, Op, gl.W_Q, O, R, dh, O, H, dhp, 0, (long long)dh * O));
    float* drres = b.take((size_t)R * Op);
    RUN(nt(dQ, (long long)H * dhp, w.WQ_t, (long long)H * dhp, drres, Op, R, O, H * dhp, 1, 0, 0, 0));
    RUN(tgmx_add_cols(drres, Op, du, Op, R, O, 1, stream));  // + the residual branch
    // rres = [x | 0 | cos(tb)]:  d tb -= sin(tb) * colsum(drres[:, time columns]);  d x = drres[:, :d]
    float* g_time_cols = b.take((size_t)T);
    RUN_ALWAYS(colsum(drres + (O - T), Op, R, T, g_time_cols, 0));
    RUN_ALWAYS(flush());  // every split reduction of this layer: two launches
    if (!dry) {
      hipLaunchKernelGGL(tgmx::time_bias_residual_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, g->tb, m->tb, g_time_cols, T);
      TGMX_CHECK_LAUNCH("tgat_backward");
    }
    if (need_dprev) RUN(tgmx_add_cols(dprev, d, drres, Op, R, d, 1, stream));
    dout = dprev;
    ld_dout = d;
  }
#undef RUN
#undef RUN_ALWAYS
  if (need_floats) *need_floats = b.off;
  return TGMX_OK;
}

int check_backward_args(const tgmx_tgat_model_t* m, const tgmx_tgat_layout_t* lay, const tgmx_tgat_hop_t* hops) {
  TGMX_REQUIRE(m && lay && hops, "tgat_backward: null pointer");
  const int L = m->num_layers;
  TGMX_REQUIRE(L >= 1 && L <= TGMX_TGAT_MAX_LAYERS, "tgat_backward: num_layers=%d outside [1, %d]", L, TGMX_TGAT_MAX_LAYERS);
  for (int l = 0; l < L; ++l) {
    const tgmx_tgat_layer_t& ly = m->layers[l];
    TGMX_REQUIRE(ly.T == m->layers[0].T && ly.H > 0 && ly.O % ly.H == 0 && ly.T <= ly.O, "tgat_backward: layer %d has an unsupported shape", l);
    TGMX_REQUIRE(lay->layers[l].probs >= 0, "tgat_backward: the forward was not run with save = 1");
  }
  return TGMX_OK;
}
}  // namespace

extern "C" size_t tgmx_tgat_backward_workspace_bytes(const tgmx_tgat_model_t* m, const tgmx_tgat_layout_t* lay, const tgmx_tgat_hop_t* hops) {
  if (check_backward_args(m, lay, hops)) return 0;
  size_t floats = 0;  // (the weight-gradient GEMMs' split partials live in the same bump area: they outlive the GEMM until the layer's flush)
  if (tgat_backward_pass(m, lay, hops, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, true, &floats)) return 0;
  return floats * sizeof(float) + 512;
}

extern "C" int tgmx_tgat_backward(const tgmx_tgat_model_t* m, const tgmx_tgat_layout_t* lay, const tgmx_tgat_hop_t* hops, const float* saved,
                                  const float* dz, int64_t ldz, const tgmx_dropout_t* drop, const tgmx_tgat_grads_t* grads, float* workspace,
                                  size_t workspace_bytes, tgmx_stream_t stream) {
  if (const int rc = check_backward_args(m, lay, hops)) return rc;
  TGMX_REQUIRE(saved && dz && grads && workspace, "tgat_backward: null pointer");
  TGMX_REQUIRE(grads->tw && grads->tb, "tgat_backward: null gradient pointer");
  for (int l = 0; l < m->num_layers; ++l) {
    const tgmx_tgat_layer_grads_t& gl = grads->layers[l];
    TGMX_REQUIRE(gl.W_Q && gl.W_KV && gl.W_O && gl.b_O && gl.ln_g && gl.ln_b && gl.fc1_w && gl.fc1_b && gl.fc2_w && gl.fc2_b,
                 "tgat_backward: null gradient pointer (layer %d)", l);
  }
  TGMX_REQUIRE(ldz >= m->layers[m->num_layers - 1].emb_out, "tgat_backward: ldz=%lld", (long long)ldz);
  const size_t need = tgmx_tgat_backward_workspace_bytes(m, lay, hops);
  TGMX_REQUIRE(need > 0 && workspace_bytes >= need, "tgat_backward: workspace too small (%zu of %zu bytes)", workspace_bytes, need);
  float* base = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const float* S = reinterpret_cast<const float*>(((uintptr_t)saved + 255) & ~(uintptr_t)255);  // the forward aligns its workspace the same way
  return tgat_backward_pass(m, lay, hops, S, dz, (long long)ldz, drop, grads, base, (hipStream_t)stream, false, nullptr);
}
