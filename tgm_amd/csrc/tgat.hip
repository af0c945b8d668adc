// TGAT / TemporalAttention forward (fp32) for gfx950.
//
// Restructured attention (derivation: oracle/tgat_fold.py): the query length is 1,
// so the per-slot W_KV projection of the reference (tgm/nn/modules/attention.py:98-101,
// 28 GFLOP per batch at the headline config) is folded onto the query / output side.
// What remains per row r (one 64-lane wave per row):
//     score[h][s] = qf[r,h,:] . z[r,s,:] * dh^-1/2      z = [nbr_x | edge_x | cos(dt w + b)]
//     A[h][:]     = softmax(mask(score[h][:]))
//     zbar[r,h,:] = sum_s A[h][s] z[r,s,:]
// -- two streaming passes over the row's [k, C] gathered features (the second one
// hits L1/L2), wave-level reductions, no GEMM: HBM-bound.  The dense contractions
// that are left ([R,O]x[O,O], [R,dh]x[dh,C], [R,C]x[C,dh], merge MLP) run on the
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) in sgemm_nt below.
#include "common.h"

#include <atomic>
#include <type_traits>

namespace tgmx {

using floatx16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

// ---------------------------------------------------------------------------
// C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]);  row-major, arbitrary leading dims,
// optional batch (grid.z) with element strides.  Exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32), operands loaded STRAIGHT INTO THE MFMA REGISTER LAYOUT:
// both A and B are K-contiguous, so lane (i = lane & 31, half = lane >> 5) loads the
// 8 consecutive k of "its" row -- A[m0+i][k0 + 8*half ..] and B[n0+i][k0 + 8*half ..]
// (two 16-byte loads each) -- and the kk-th MFMA of a 16-wide k step consumes
// k0+kk from the lower half-wave and k0+8+kk from the upper one.  No LDS, no
// barriers; a wave owns a 32 x 64 output tile (two accumulators), four waves stack
// along M.  The next step's operands are loaded before the current step's 16 MFMAs.
// ---------------------------------------------------------------------------
struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long long lda, ldb, ldc, sA, sB, sC;
  long long M;
  int N, K, relu;
};

// One k-step covers 2 * kGemmK columns of K: lanes 0-31 hold the first kGemmK of their row, lanes 32-63 the next
// kGemmK (any pairing of k with the two k-slots of the 32x32x2 MFMA is valid as long as A and B agree).
// kGemmK = 8 measured best on this path's skinny GEMMs (16 / 32 cost occupancy: 216+ VGPRs, one wave per SIMD).

template <bool VEC, int kGemmK>
__device__ __forceinline__ void load_kn(const float* __restrict__ row, int kb, int K, float (&v)[kGemmK]) {
#pragma unroll
  for (int c = 0; c < kGemmK; c += 4) {
    if (VEC && kb + c + 4 <= K) {
      const float4 x = *reinterpret_cast<const float4*>(row + kb + c);
      v[c] = x.x; v[c + 1] = x.y; v[c + 2] = x.z; v[c + 3] = x.w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[c + u] = kb + c + u < K ? row[kb + c + u] : 0.f;
    }
  }
}

// C[M, N] = A[M, K] . B[N, K]^T (+ bias, relu), both operands K-contiguous and loaded straight into the MFMA
// register layout (no LDS staging).  SPLIT = false: the 4 waves of a block own 4 row tiles (128 rows).
// SPLIT = true (few tiles): the 4 waves share ONE 32 x 64 tile, wave w takes k-steps w, w + 4, ... and the
// partial tiles meet in LDS (fixed summation order: deterministic).
// EXT: the K-split reduction's 32 KB live in LDS the caller provides (a kernel that also instantiates the LDS-staged body: one buffer for both)
template <bool AV, bool BV, bool SPLIT, int kGemmK, bool EXT = false>
__device__ __forceinline__ void sgemm_nt_body(const GemmArgs& g, const unsigned bx, const unsigned by, const unsigned bz, float* ext = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;
  const long long m0 = SPLIT ? (long long)bx * 32 : (long long)bx * 128 + wave * 32;
  if (!SPLIT && m0 >= g.M) return;  // whole wave out of range (no block-level sync on this path)
  const int n0 = by * 64;
  const float* __restrict__ A = g.A + (long long)bz * g.sA;
  const float* __restrict__ B = g.B + (long long)bz * g.sB;
  float* __restrict__ C = g.C + (long long)bz * g.sC;
  // rows / columns past the edge are clamped for the loads; their results are never stored
  const long long ra = m0 + i < g.M ? m0 + i : g.M - 1;
  const int c0 = n0 + i < g.N ? n0 + i : g.N - 1;
  const int c1 = n0 + 32 + i < g.N ? n0 + 32 + i : g.N - 1;
  const float* __restrict__ pa = A + ra * g.lda;
  const float* __restrict__ pb0 = B + (long long)c0 * g.ldb;
  const float* __restrict__ pb1 = B + (long long)c1 * g.ldb;

  floatx16 acc0 = {0}, acc1 = {0};
  float a[kGemmK], b0[kGemmK], b1[kGemmK], an[kGemmK], b0n[kGemmK], b1n[kGemmK];
  constexpr int kStep = 2 * kGemmK;
  const int stride = SPLIT ? 4 * kStep : kStep;
  int k0 = SPLIT ? wave * kStep : 0;
  if (k0 < g.K) {
    load_kn<AV, kGemmK>(pa, k0 + half * kGemmK, g.K, a);
    load_kn<BV, kGemmK>(pb0, k0 + half * kGemmK, g.K, b0);
    load_kn<BV, kGemmK>(pb1, k0 + half * kGemmK, g.K, b1);
  }
  for (; k0 < g.K; k0 += stride) {
    const bool more = k0 + stride < g.K;
    if (more) {
      const int kb = k0 + stride + half * kGemmK;
      load_kn<AV, kGemmK>(pa, kb, g.K, an);
      load_kn<BV, kGemmK>(pb0, kb, g.K, b0n);
      load_kn<BV, kGemmK>(pb1, kb, g.K, b1n);
    }
#pragma unroll
    for (int kk = 0; kk < kGemmK; ++kk) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b0[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b1[kk], acc1, 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < kGemmK; ++u) {
        a[u] = an[u];
        b0[u] = b0n[u];
        b1[u] = b1n[u];
      }
    }
  }
  if (SPLIT) {
    float (*part)[32][kWave];  // [wave][acc register (0-15: acc0, 16-31: acc1)][lane]
    if constexpr (EXT) {
      part = reinterpret_cast<float (*)[32][kWave]>(ext);
    } else {
      __shared__ float part_own[4][32][kWave];
      part = part_own;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      part[wave][r][lane] = acc0[r];
      part[wave][16 + r][lane] = acc1[r];
    }
    __syncthreads();
    // wave w finishes registers 8w .. 8w+7 of every lane
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int rr = wave * 8 + q;
      float v = part[0][rr][lane] + part[1][rr][lane] + part[2][rr][lane] + part[3][rr][lane];
      const int r = rr & 15, t = rr >> 4;
      const long long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int col = n0 + t * 32 + i;
      if (row < g.M && col < g.N) {
        if (g.bias) v += g.bias[(long long)bz * g.N + col];  // batch b adds row b of a [batch, N] bias
        if (g.relu) v = v > 0.f ? v : 0.f;
        C[row * g.ldc + col] = v;
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (row >= g.M) continue;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = n0 + t * 32 + i;
      if (col < g.N) {
        float v = t ? acc1[r] : acc0[r];
        if (g.bias) v += g.bias[(long long)bz * g.N + col];  // batch b adds row b of a [batch, N] bias
        if (g.relu) v = v > 0.f ? v : 0.f;
        C[row * g.ldc + col] = v;
      }
    }
  }
}

template <bool AV, bool BV, bool SPLIT, int kGemmK>
__global__ __launch_bounds__(256) void sgemm_nt_kernel(const GemmArgs g) {
  sgemm_nt_body<AV, BV, SPLIT, kGemmK>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// TWO independent GEMMs in one launch (the K-split kernel's body, unchanged: each problem's result is bit for bit what its own launch gives).
// Workgroups [0, tiles0) of the x dimension belong to problem 0, the rest to problem 1; the y / z extents are the larger of the two
// problems' (a workgroup outside its problem's extent returns at once).  For pairs that do not depend on each other and are each too
// small to fill the chip -- the GRU's gi / gh, TransformerConv's node projections / edge projection (cfg 3: four launches of ~14 us -> two).
struct GemmPairArgs {
  GemmArgs g[2];
  unsigned tiles0, ny[2], nz[2];
  int lds[2];  // (mixed pairs) the problem takes the LDS-staged body
};

__global__ __launch_bounds__(256) void sgemm_nt_pair_kernel(const GemmPairArgs p) {
  const unsigned which = blockIdx.x >= p.tiles0;
  if (blockIdx.y >= p.ny[which] || blockIdx.z >= p.nz[which]) return;
  if (which) sgemm_nt_body<true, true, true, 8>(p.g[1], blockIdx.x - p.tiles0, blockIdx.y, blockIdx.z);
  else sgemm_nt_body<true, true, true, 8>(p.g[0], blockIdx.x, blockIdx.y, blockIdx.z);
}

// The MANY-ROW variant (M >= 6144, see gemm_takes_lds): both operands STAGED THROUGH LDS in whole 128-byte lines.
// Why: the direct kernel above takes 32 bytes of every 128-byte line per lane and k-step straight from L2 into the
// MFMA layout; with 4-5 workgroups per CU in flight the lines do not survive in the 32 KB L1 until their other quarters are wanted, and the
// launch runs at 0.22-0.34 of the fp32-MFMA peak however it is tiled (12 600 x 172 x 172: 23 us for 4.8 us of MFMAs, ~4 us per layer of
// 256 workgroups).  Here a workgroup owns a 64 x 64 block; per 32-wide k-chunk its 256 threads fetch the block's 64 + 64 rows as 128-byte
// row segments (eight lanes x 16 bytes per row: every line whole, once), park them in registers while the current chunk's MFMAs run, and
// store them into the other LDS buffer (rows padded to 36 floats: conflict-free 16-byte reads).  Wave (wr, wc) multiplies rows
// [32 wr, 32 wr + 32) with columns [32 wc, 32 wc + 32): lane (i, half) reads k-set [16 half, 16 half + 16) of its A row and its B row (four
// 16-byte LDS reads each) for the chunk's 16 MFMAs -- the same pairing of k with the two k-slots as above.  One barrier per chunk.
// Summation order: chunk by chunk in ascending k, no cross-wave reduction (differs from the K-split kernel at rounding level).
constexpr int kLdsBM = 64, kLdsBN = 64, kLdsBK = 32, kLdsLd = kLdsBK + 4;
constexpr int kLdsFloats = 2 * (kLdsBM + kLdsBN) * kLdsLd;  // two buffers of (A block | B block): 36 864 bytes

// one 16-byte piece of a row's k-chunk: zeros past K (K need not be a multiple of 4; the padding of a padded row is not trusted)
// (VEC = false: rows that are not 16-byte aligned -- column-sliced views -- are read float by float: same values, same sums)
template <bool VEC>
__device__ __forceinline__ float4 lds_gemm_piece(const float* __restrict__ row, int k, int K) {
  if (VEC && k + 4 <= K) return *reinterpret_cast<const float4*>(row + k);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < K) v.x = row[k];
  if (k + 1 < K) v.y = row[k + 1];
  if (k + 2 < K) v.z = row[k + 2];
  if (k + 3 < K) v.w = row[k + 3];
  return v;
}

template <bool AV, bool BV>
__device__ __forceinline__ void sgemm_nt_lds_body(const GemmArgs& g, const unsigned bx, const unsigned by, const unsigned bz, float* __restrict__ smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const long long m0 = (long long)bx * kLdsBM;
  const int n0 = by * kLdsBN;
  const float* __restrict__ A = g.A + (long long)bz * g.sA;
  const float* __restrict__ B = g.B + (long long)bz * g.sB;
  float* __restrict__ C = g.C + (long long)bz * g.sC;
  // staging: thread t moves piece (t & 7) of rows (t >> 3) and (t >> 3) + 32 of the A block and of the B block (rows past the edge are
  // clamped: their products are never stored)
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  const long long ra0 = m0 + sr < g.M ? m0 + sr : g.M - 1, ra1 = m0 + sr + 32 < g.M ? m0 + sr + 32 : g.M - 1;
  const int rb0 = n0 + sr < g.N ? n0 + sr : g.N - 1, rb1 = n0 + sr + 32 < g.N ? n0 + sr + 32 : g.N - 1;
  const float* __restrict__ pa0 = A + ra0 * g.lda;
  const float* __restrict__ pa1 = A + ra1 * g.lda;
  const float* __restrict__ pb0 = B + (long long)rb0 * g.ldb;
  const float* __restrict__ pb1 = B + (long long)rb1 * g.ldb;
  auto As = [&](int buf) { return smem + buf * (kLdsBM + kLdsBN) * kLdsLd; };
  auto Bs = [&](int buf) { return smem + buf * (kLdsBM + kLdsBN) * kLdsLd + kLdsBM * kLdsLd; };
  const int nchunks = (g.K + kLdsBK - 1) / kLdsBK;
  float4 v[4];  // the thread's four staged pieces: rows sr and sr + 32 of the A block and of the B block
  auto request = [&](int chunk) __attribute__((always_inline)) {
    const int k = chunk * kLdsBK + sc;
    v[0] = lds_gemm_piece<AV>(pa0, k, g.K); v[1] = lds_gemm_piece<AV>(pa1, k, g.K);
    v[2] = lds_gemm_piece<BV>(pb0, k, g.K); v[3] = lds_gemm_piece<BV>(pb1, k, g.K);
  };
  auto park = [&](int buf) __attribute__((always_inline)) {
    *reinterpret_cast<float4*>(As(buf) + sr * kLdsLd + sc) = v[0];
    *reinterpret_cast<float4*>(As(buf) + (sr + 32) * kLdsLd + sc) = v[1];
    *reinterpret_cast<float4*>(Bs(buf) + sr * kLdsLd + sc) = v[2];
    *reinterpret_cast<float4*>(Bs(buf) + (sr + 32) * kLdsLd + sc) = v[3];
  };
  request(0);
  park(0);
  __syncthreads();
  floatx16 acc = {0};
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) request(c + 1);  // the next chunk's lines: in flight during this chunk's MFMAs
    const float4* __restrict__ ar = reinterpret_cast<const float4*>(As(c & 1) + (wr * 32 + i) * kLdsLd + half * 16);
    const float4* __restrict__ br = reinterpret_cast<const float4*>(Bs(c & 1) + (wc * 32 + i) * kLdsLd + half * 16);
    float4 a4[4], b4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a4[u] = ar[u]; b4[u] = br[u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u].x, b4[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u].y, b4[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u].z, b4[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u].w, b4[u].w, acc, 0, 0, 0);
    }
    if (more) {  // (the other buffer was last read in iteration c - 1, behind that iteration's barrier)
      park((c + 1) & 1);
      __syncthreads();
    }
  }
  const int col = n0 + wc * 32 + i;
  if (col < g.N) {
    const float bias = g.bias ? g.bias[(long long)bz * g.N + col] : 0.f;  // batch b adds row b of a [batch, N] bias
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < g.M) {
        float v = acc[r];
        if (g.bias) v += bias;
        if (g.relu) v = v > 0.f ? v : 0.f;
        C[row * g.ldc + col] = v;
      }
    }
  }
}

template <bool AV, bool BV>
__global__ __launch_bounds__(256) void sgemm_nt_lds_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[kLdsFloats];
  sgemm_nt_lds_body<AV, BV>(g, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// two independent problems in one launch of the kernel above (see sgemm_nt_pair_kernel)
__global__ __launch_bounds__(256) void sgemm_nt_lds_pair_kernel(const GemmPairArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[kLdsFloats];
  const unsigned which = blockIdx.x >= p.tiles0;
  if (blockIdx.y >= p.ny[which] || blockIdx.z >= p.nz[which]) return;
  sgemm_nt_lds_body<true, true>(p.g[which], which ? blockIdx.x - p.tiles0 : blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// ... and a pair whose problems take DIFFERENT kernels alone (cfg 3's projections: ~5 k unique nodes beside ~13 k edges): each through
// its own body, unchanged, over one LDS buffer -- still one launch, and each result bit for bit what its own launch gives
__global__ __launch_bounds__(256) void sgemm_nt_mixed_pair_kernel(const GemmPairArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[kLdsFloats];
  static_assert(kLdsFloats >= 4 * 32 * kWave, "the K-split reduction fits the staging buffer");
  const unsigned which = blockIdx.x >= p.tiles0;
  if (blockIdx.y >= p.ny[which] || blockIdx.z >= p.nz[which]) return;
  const unsigned bx = which ? blockIdx.x - p.tiles0 : blockIdx.x;
  if (p.lds[which]) sgemm_nt_lds_body<true, true>(p.g[which], bx, blockIdx.y, blockIdx.z, smem);
  else sgemm_nt_body<true, true, true, 8, true>(p.g[which], bx, blockIdx.y, blockIdx.z, smem);
}

// The FEW-ROW variant (M <= 2048: the 600-row layer of the headline forward, whose five GEMMs were 41 us of a 185 us forward at ~8 us
// each for 0.03-0.2 GFLOP).  Such a launch is one dependent chain -- launch, operand latency, MFMAs, reduction, store -- so the
// kernel is built to make that chain SHORT instead of wide: a workgroup owns one 32 x 32 output block, its 8 waves split K into
// contiguous slices of STEPS 16-wide steps, and a wave issues EVERY load of its slice before its first MFMA (<= 16 float4 per lane:
// one memory latency per launch, where sgemm_nt_kernel<SPLIT> pays one per k-step: 4-7 of them).  The 8 partial blocks meet in LDS
// in wave order (fixed: deterministic).  32-column blocks instead of 64 double the workgroups (114 for a 600 x 172 output).
template <bool AV, bool BV, int STEPS>
__global__ __launch_bounds__(512) void sgemm_nt_small_kernel(const GemmArgs g) {
  constexpr int kW = 8;
  __shared__ float part[kW][16][kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;
  const long long m0 = (long long)blockIdx.x * 32;
  const int n0 = blockIdx.y * 32;
  const float* __restrict__ A = g.A + (long long)blockIdx.z * g.sA;
  const float* __restrict__ B = g.B + (long long)blockIdx.z * g.sB;
  float* __restrict__ C = g.C + (long long)blockIdx.z * g.sC;
  const long long ra = m0 + i < g.M ? m0 + i : g.M - 1;  // clamped for the loads; never stored
  const int cb = n0 + i < g.N ? n0 + i : g.N - 1;
  const float* __restrict__ pa = A + ra * g.lda;
  const float* __restrict__ pb = B + (long long)cb * g.ldb;
  const int kbeg = wave * STEPS * 16;
  float a[STEPS][8], b[STEPS][8];
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int k0 = kbeg + st * 16;
    if (k0 < g.K) {
      load_kn<AV, 8>(pa, k0 + half * 8, g.K, a[st]);
      load_kn<BV, 8>(pb, k0 + half * 8, g.K, b[st]);
    }
  }
  floatx16 acc = {0};
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    if (kbeg + st * 16 < g.K) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][kk], b[st][kk], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
  __syncthreads();
  // wave w finishes accumulator registers 2w and 2w + 1 of every lane
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = wave * 2 + q;
    float v = part[0][r][lane];
#pragma unroll
    for (int w = 1; w < kW; ++w) v += part[w][r][lane];
    const long long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const int col = n0 + i;
    if (row < g.M && col < g.N) {
      if (g.bias) v += g.bias[(long long)blockIdx.z * g.N + col];
      if (g.relu) v = v > 0.f ? v : 0.f;
      C[row * g.ldc + col] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// out[i, :] = table[idx[i] (negative wraps, like Python indexing), :]
// (tgat.py:128-130: pad id -1 reads the LAST row of node_x)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, long long rows, int dim,
                                                          const int32_t* __restrict__ idx, long long n,
                                                          float* out, long long ldo) {
  const long long total = n * dim;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long i = e / dim;
    const int c = (int)(e - i * dim);
    long long r = idx[i];
    if (r < 0) r += rows;
    out[i * ldo + c] = table[r * dim + c];
  }
}

// the leaf features of every level of the hop tree in one launch: out rows [end[i-1], end[i]) = table[idx_i[...]]
struct LeafGatherArgs {
  const int32_t* idx[TGMX_TGAT_MAX_LAYERS + 1];
  long long end[TGMX_TGAT_MAX_LAYERS + 1];  // exclusive end row of level i in the output
  const float* table;
  float* out;
  long long rows;
  int levels, dim;
};

__global__ __launch_bounds__(256) void gather_leaves_kernel(const LeafGatherArgs g) {
  const long long total = g.end[g.levels - 1] * g.dim;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long i = e / g.dim;
    const int c = (int)(e - i * g.dim);
    const int32_t* idx = g.idx[0];
    long long base = 0;
#pragma unroll
    for (int l = 1; l <= TGMX_TGAT_MAX_LAYERS; ++l) {
      if (l < g.levels && i >= g.end[l - 1]) {
        idx = g.idx[l];
        base = g.end[l - 1];
      }
    }
    long long r = idx[i - base];
    if (r < 0) r += g.rows;  // pad id -1 reads the LAST row of node_x (tgat.py:128-130)
    g.out[e] = g.table[r * g.dim + c];
  }
}

// Compact rows: behind tgmx_pair_dedup of every level >= 1.  Part A, one item per ORIGINAL row of a deduplicated level: rep[r] =
// cidx[owner[r]], the compact row that stands for it (what the level above indexes its neighbor features with).  Part B, one item
// per compact row and column: the leaf features z0[level i][c] = node_x[id of the row compact row c stands for] (pad id -1 reads the
// LAST row, tgat.py:128-130); level 0 is not deduplicated.  Items past a level's device-side count do nothing.
struct CompactGatherArgs {
  int levels;                                         // levels 0 .. levels - 1 are row batches (the deepest level is read by node id)
  long long rows[TGMX_TGAT_MAX_LAYERS + 1];           // original rows per level
  long long off[TGMX_TGAT_MAX_LAYERS + 1];            // first row of the level in z0
  const int32_t* ids[TGMX_TGAT_MAX_LAYERS + 1];       // level 0: the seeds; level i: hop i-1's neighbor ids
  const int32_t* uniq[TGMX_TGAT_MAX_LAYERS + 1];
  const int32_t* owner[TGMX_TGAT_MAX_LAYERS + 1];
  const int32_t* cidx[TGMX_TGAT_MAX_LAYERS + 1];
  int32_t* rep[TGMX_TGAT_MAX_LAYERS + 1];
  const int32_t* count[TGMX_TGAT_MAX_LAYERS + 1];
  const float* table;
  float* out;
  long long num_nodes, itemsA, itemsB;
  int dim;
};
__global__ __launch_bounds__(256) void gather_compact_kernel(const CompactGatherArgs g) {
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < g.itemsA + g.itemsB; e += step) {
    if (e < g.itemsA) {
      long long r = e;
      int lv = 1;
#pragma unroll
      for (int l = 1; l < TGMX_TGAT_MAX_LAYERS; ++l)
        if (l < g.levels - 1 && r >= g.rows[lv] && lv == l) { r -= g.rows[lv]; lv = l + 1; }
      g.rep[lv][r] = g.cidx[lv][g.owner[lv][r]];
    } else {
      const long long f = e - g.itemsA;
      long long i = f / g.dim;
      const int c = (int)(f - i * g.dim);
      int lv = 0;
#pragma unroll
      for (int l = 0; l < TGMX_TGAT_MAX_LAYERS; ++l)
        if (l < g.levels - 1 && i >= g.rows[lv] && lv == l) { i -= g.rows[lv]; lv = l + 1; }
      long long orig = i;
      if (lv > 0) {
        if (i >= (long long)*g.count[lv]) continue;
        orig = g.uniq[lv][i];
      }
      long long node = g.ids[lv][orig];
      if (node < 0) node += g.num_nodes;
      g.out[(g.off[lv] + i) * g.dim + c] = g.table[node * g.dim + c];
    }
  }
}

// Rres[r] = [x[r, :d] | 0 (pad) | cos(tb)]  -- the residual == query input (attention.py:93-95);
// Time2Vec of the zero vector is cos(fma(0, w, b)) = cos(b).
__global__ __launch_bounds__(256) void tgat_rres_kernel(const float* __restrict__ x, long long ldx, int d,
                                                        const float* __restrict__ tb, const float* __restrict__ tfeat,
                                                        int T, int O, long long R, float* __restrict__ out, long long ldo) {
  // columns [O, ldo) of a padded row are zero-filled
  const long long total = R * ldo;
  const long long step = (long long)gridDim.x * blockDim.x;
  const int t0 = O - T;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long r = e / ldo;
    const int c = (int)(e - r * ldo);
    float v = 0.f;
    if (c < d) v = x[r * ldx + c];
    else if (c >= t0 && c < O) v = tfeat ? tfeat[r * T + (c - t0)] : cos_t2v(tb[c - t0]);
    out[e] = v;
  }
}

// out[i, t] = cos(fma(float(x[i]), w[t], b[t]))   (Time2Vec.forward, time_encoding.py:22-24)
template <typename TI>
__global__ __launch_bounds__(256) void time2vec_kernel(const TI* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, int T, long long n,
                                                       float* __restrict__ out) {
  const long long total = n * T;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long i = e / T;
    const int t = (int)(e - i * T);
    out[e] = cos_t2v(__fmaf_rn((float)x[i], w[t], b[t]));
  }
}

// out[r, :O] = LayerNorm(y[r] + res[r]) * gamma + beta ; out[r, O:O+d0] = z0[r]   (one wave per row)
// res == nullptr: the residual is rebuilt here, [x[r, :d] | 0 | cos(tb)] (what tgat_rres_kernel would have written: same values)
__global__ __launch_bounds__(256) void ln_residual_concat_kernel(const float* __restrict__ y, const float* __restrict__ res,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int O, float eps,
                                                                 const float* __restrict__ z0, int d0, long long R,
                                                                 float* __restrict__ out, long long ldy, long long ldr,
                                                                 long long ldo, const float* __restrict__ x, long long ldx, int d,
                                                                 const float* __restrict__ tb, int T) {
  const int lane = lane_id();
  const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* yr = y + r * ldy;
  const float* rr = res ? res + r * ldr : nullptr;
  const float* xr = x + r * ldx;
  const int t0 = O - T;
  auto resid = [&](int c) -> float { return rr ? rr[c] : (c < d ? xr[c] : (c >= t0 ? cos_t2v(tb[c - t0]) : 0.f)); };
  float* orow = out + r * ldo;
  if (O <= 8 * kWave) {  // the row's y + residual held in registers: one read, one Time2Vec per column
    float t[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * kWave;
      t[i] = c < O ? yr[c] + resid(c) : 0.f;
      s += t[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)O;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float u = lane + i * kWave < O ? t[i] - mean : 0.f;
      v += u * u;
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / (float)O + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * kWave;
      if (c < O) orow[c] = (t[i] - mean) * rstd * gamma[c] + beta[c];
    }
    for (int c = lane; c < d0; c += kWave) orow[O + c] = z0[r * d0 + c];
    for (int c = O + d0 + lane; c < ldo; c += kWave) orow[c] = 0.f;  // padding of a padded row
    return;
  }
  float s = 0.f;
  for (int c = lane; c < O; c += kWave) s += yr[c] + resid(c);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)O;
  float v = 0.f;
  for (int c = lane; c < O; c += kWave) {
    const float t = yr[c] + resid(c) - mean;
    v += t * t;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = 1.0f / sqrtf(v / (float)O + eps);
  for (int c = lane; c < O; c += kWave) orow[c] = (yr[c] + resid(c) - mean) * rstd * gamma[c] + beta[c];
  for (int c = lane; c < d0; c += kWave) orow[O + c] = z0[r * d0 + c];
  for (int c = O + d0 + lane; c < ldo; c += kWave) orow[c] = 0.f;  // padding of a padded row
}

// ---------------------------------------------------------------------------
// Everything behind the attention reduce of one layer, for a tile of 32 rows, in ONE kernel:
//   oattn = [zbar_h . W_V,h^T]_h   ->   y = oattn . W_O^T + b_O   ->   LayerNorm(y + [x | 0 | cos(tb)])
//   ->   cat [ . | z0 ]   ->   relu(fc1)   ->   fc2                     (attention.py:119-127, tgat.py:36-38)
// Every stage is row-local, so the intermediates never leave LDS: five launches (each ~5 us of fixed cost at these
// sizes) and four HBM round trips of [R, ~O] activations collapse into one.  A stage is a 32 x N x K GEMM: the A
// operand comes from LDS (row stride = 4 mod 32 floats: conflict-free 16-byte reads) or, for the first stage,
// straight from global memory; the weights (B, K-contiguous rows) are read from L2 in the MFMA register layout;
// the 4 waves split the 32-column blocks of N.
// ---------------------------------------------------------------------------
// Which rows of a launch are LIVE.  Compact rows (tgmx_tgat_hop_t.seed_keyed): every level of the hop tree keeps its slice
// [begin[i], begin[i + 1]) of a layer's row space, but only the first *live[i] rows of a deduplicated level exist -- a device-side
// count, so launches are sized for the worst case and the workgroups / waves past the count leave at once.  n == 0: all rows live.
struct RowSegs {
  int n;
  long long begin[TGMX_TGAT_MAX_LAYERS + 1];
  const int32_t* live[TGMX_TGAT_MAX_LAYERS];
};
// any live row in [r0, r1)?  (workgroup- / wave-uniform arguments: scalar code)
__device__ __forceinline__ bool rows_live(const RowSegs& s, long long r0, long long r1) {
  if (s.n == 0) return true;
  bool any = false;
#pragma unroll
  for (int i = 0; i < TGMX_TGAT_MAX_LAYERS; ++i) {
    if (i < s.n) {
      const long long lo = r0 > s.begin[i] ? r0 : s.begin[i];
      const long long hi = r1 < s.begin[i + 1] ? r1 : s.begin[i + 1];
      if (lo < hi && (!s.live[i] || lo - s.begin[i] < (long long)*s.live[i])) any = true;
    }
  }
  return any;
}

struct ChainArgs {
  RowSegs segs;
  const float* zbar;  // [R, H, Cp]
  const float* x;     // [R, >= d] layer input rows (the residual's feature part)
  const float* tb;    // [T] Time2Vec bias: the residual's time part is cos(tb) (attention.py:93-95)
  const float* z0;    // [R, d0] skip features for the merge layer
  const float *W_V, *W_O, *b_O, *ln_g, *ln_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  const float* W_Vc;  // chain64, H == 2: both heads' W_V stacked (per head zero-padded to 16-row blocks) and tiled as ONE matrix; NULL: per head
  float* out;         // [R, ldo]
  long long R, ld_zbar, ldx, ldo;
  int d, T, d0, O, H, C, emb, emb_out;
  int Cp, Op, Kc, Ep;  // row strides of the padded weight copies
  int LD;              // LDS row stride in floats
  float eps;
};

template <bool A_LDS>
__device__ __forceinline__ void chain_load_a(const float* __restrict__ row, int kb, int K, float (&v)[8]) {
  if (kb + 8 <= K) {
    const float4 x = *reinterpret_cast<const float4*>(row + kb);
    const float4 y = *reinterpret_cast<const float4*>(row + kb + 4);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = kb + u < K ? row[kb + u] : 0.f;  // LDS columns >= K may hold an older stage
  }
}

__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[8]) {
  const float4 x = *reinterpret_cast<const float4*>(p);
  const float4 y = *reinterpret_cast<const float4*>(p + 4);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
}

// one 32 x 32 output block: out[:, col_off + n0 .. +32) = A[32, K] . B[n0 .. n0+32, K]^T (+ bias, relu).
// The weights come from L2 (~1 us away) and a 16-wide k-step is only 8 MFMAs (0.2 us), so the k-loop runs in
// groups of 4 steps with 4 register buffers: straight-line code, every buffer is refilled for the next group right
// after its MFMAs issue (no branch between a load and its use, so the compiler's waitcnt stays exact).  The
// < 4 leftover steps (and the ragged tail of K) are loaded up front with guards and consumed last.
template <bool A_LDS, bool OUT_LDS>
__device__ __forceinline__ void chain_block(const float* __restrict__ A, long long lda, int rows_valid,
                                            const float* __restrict__ B, long long ldb, int N, int n0, int K,
                                            const float* __restrict__ bias, int relu, float* __restrict__ out,
                                            long long ldo, int col_off, int lane) {
  const int i = lane & 31, half = lane >> 5;
  const int cb = n0 + i < N ? n0 + i : N - 1;
  const float* __restrict__ pb = B + (long long)cb * ldb + half * 8;
  const float* __restrict__ pa = A + (long long)((A_LDS || i < rows_valid) ? i : rows_valid - 1) * lda + half * 8;
  floatx16 acc = {0};
  constexpr int GA = A_LDS ? 1 : 8;  // a global A operand rides along with B; an LDS one is read at use
  const int nfull = K > 0 ? K / 16 : 0, ngrp = nfull / 4;
  const int rem0 = ngrp * 4;  // first leftover step
  float r0[8], r1[8], r2[8], r3[8], s0[GA], s1[GA], s2[GA], s3[GA];
  auto tail_fetch = [&](float (&bd)[8], float (&ad)[GA], int st) {
    const int kb = st * 16 + half * 8;  // kb is relative to pa / pb, which already include half * 8
    if (st * 16 < K) {
      load_kn<true, 8>(pb - half * 8, kb, K, bd);
      if constexpr (!A_LDS) chain_load_a<false>(pa - half * 8, kb, K, ad);
    }
  };
  tail_fetch(r0, s0, rem0);
  tail_fetch(r1, s1, rem0 + 1);
  tail_fetch(r2, s2, rem0 + 2);
  tail_fetch(r3, s3, rem0 + 3);
  auto mfma8 = [&](const float (&a)[8], const float (&b)[8]) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
  };
  if (ngrp > 0) {
    float b0[8], b1[8], b2[8], b3[8], a0[8], a1[8], a2[8], a3[8];
    load8(pb, b0);
    load8(pb + 16, b1);
    load8(pb + 32, b2);
    load8(pb + 48, b3);
    if constexpr (!A_LDS) {
      load8(pa, a0);
      load8(pa + 16, a1);
      load8(pa + 32, a2);
      load8(pa + 48, a3);
    }
    for (int gi = 0; gi < ngrp; ++gi) {
      const int cur = gi * 64;
      const int nxt = (gi + 1 < ngrp ? gi + 1 : gi) * 64;  // the last group refills itself: harmless, keeps the body branch-free
      if constexpr (A_LDS) load8(pa + cur, a0);
      mfma8(a0, b0);
      load8(pb + nxt, b0);
      if constexpr (!A_LDS) load8(pa + nxt, a0);
      if constexpr (A_LDS) load8(pa + cur + 16, a1);
      mfma8(a1, b1);
      load8(pb + nxt + 16, b1);
      if constexpr (!A_LDS) load8(pa + nxt + 16, a1);
      if constexpr (A_LDS) load8(pa + cur + 32, a2);
      mfma8(a2, b2);
      load8(pb + nxt + 32, b2);
      if constexpr (!A_LDS) load8(pa + nxt + 32, a2);
      if constexpr (A_LDS) load8(pa + cur + 48, a3);
      mfma8(a3, b3);
      load8(pb + nxt + 48, b3);
      if constexpr (!A_LDS) load8(pa + nxt + 48, a3);
    }
  }
  auto tail_step = [&](const float (&bd)[8], const float (&ad)[GA], int st) {
    if (st * 16 < K) {
      if constexpr (A_LDS) {
        float a[8];
        chain_load_a<true>(pa - half * 8, st * 16 + half * 8, K, a);
        mfma8(a, bd);
      } else {
        mfma8(ad, bd);
      }
    }
  };
  tail_step(r0, s0, rem0);
  tail_step(r1, s1, rem0 + 1);
  tail_step(r2, s2, rem0 + 2);
  tail_step(r3, s3, rem0 + 3);
  const int col = n0 + i;
  if (col < N) {
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = acc[r] + bv;
      if (relu) v = v > 0.f ? v : 0.f;
      if (OUT_LDS || row < rows_valid) out[(long long)row * ldo + col_off + col] = v;
    }
  }
}

constexpr int kChainWaves = 4;
constexpr int kChainThreads = kChainWaves * kWave;

__global__ __launch_bounds__(kChainThreads) void tgat_post_chain_kernel(const ChainArgs g) {
  // LDS: two [32, LD] activation buffers, then the row-constant vectors and this tile's residual / skip inputs,
  // all staged up front so that no stage waits on a dependent global load
  extern __shared__ __attribute__((aligned(16))) float chain_lds[];
  const int O = g.O, dh = O / g.H, LD = g.LD;
  float* buf0 = chain_lds;
  float* buf1 = buf0 + 32 * LD;
  float* ct = buf1 + 32 * LD;  // [T]  cos(tb)
  float* lg = ct + g.T;        // [O]  LayerNorm weight
  float* lb = lg + O;          // [O]  LayerNorm bias
  float* xs = lb + O;          // [32, d]  residual feature part
  float* zs = xs + 32 * g.d;   // [32, d0] skip features
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long m0 = (long long)blockIdx.x * 32;
  const int rows = (g.R - m0) < 32 ? (int)(g.R - m0) : 32;
  if (!rows_live(g.segs, m0, m0 + rows)) return;  // compact rows: nothing of this tile exists
  for (int t = tid; t < g.T; t += kChainThreads) ct[t] = cos_t2v(g.tb[t]);
  for (int c = tid; c < O; c += kChainThreads) {
    lg[c] = g.ln_g[c];
    lb[c] = g.ln_b[c];
  }
  for (int e = tid; e < rows * g.d; e += kChainThreads) {
    const int r = e / g.d, c = e - r * g.d;
    xs[e] = g.x[(m0 + r) * g.ldx + c];
  }
  for (int e = tid; e < rows * g.d0; e += kChainThreads) zs[e] = g.z0[m0 * g.d0 + e];

  // stage 1: oattn (buf0) -- per head, zbar rows straight from global.  K = C is the long dimension here and there
  // are only H * ceil(dh / 32) column blocks, so two waves share a block: the upper half of K lands in buf1 and is
  // added by the owner of the lower half (fixed order: deterministic).
  const int nb_h = (dh + 31) / 32;
  const int nblk = g.H * nb_h;
  const int Kh = ((g.C + 31) / 32) * 16;  // lower half of K, a multiple of the 16-wide k-step
  for (int idx = wave; idx < 2 * nblk; idx += kChainWaves) {
    const int part = idx / nblk, blk = idx - part * nblk;
    const int h = blk / nb_h, nb = blk - h * nb_h;
    const float* A = g.zbar + m0 * g.ld_zbar + (long long)h * g.Cp + part * Kh;
    const float* B = g.W_V + (long long)h * dh * g.Cp + part * Kh;
    const int Kp = part ? g.C - Kh : (Kh < g.C ? Kh : g.C);
    chain_block<false, true>(A, g.ld_zbar, rows, B, g.Cp, dh, nb * 32, Kp, nullptr, 0, part ? buf1 : buf0, LD, h * dh, lane);
  }
  __syncthreads();
  for (int e = tid; e < 32 * O; e += kChainThreads) {
    const int r = e / O, c = e - r * O;
    buf0[r * LD + c] += buf1[r * LD + c];
  }
  __syncthreads();
  // stage 2: y (buf1) = oattn . W_O^T + b_O
  for (int nb = wave; nb < (O + 31) / 32; nb += kChainWaves)
    chain_block<true, true>(buf0, LD, 32, g.W_O, g.Op, O, nb * 32, O, g.b_O, 0, buf1, LD, 0, lane);
  __syncthreads();
  // stage 3: cat (buf0) = [LayerNorm(y + residual) | z0]; same arithmetic as ln_residual_concat_kernel
  const int t0 = O - g.T;
  for (int r = wave; r < rows; r += kChainWaves) {
    const float* yr = buf1 + r * LD;
    const float* xr = xs + r * g.d;
    auto res = [&](int c) -> float { return c < g.d ? xr[c] : (c >= t0 ? ct[c - t0] : 0.f); };
    float s = 0.f;
    for (int c = lane; c < O; c += kWave) s += yr[c] + res(c);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)O;
    float v = 0.f;
    for (int c = lane; c < O; c += kWave) {
      const float t = yr[c] + res(c) - mean;
      v += t * t;
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v / (float)O + g.eps);
    float* orow = buf0 + r * LD;
    for (int c = lane; c < O; c += kWave) orow[c] = (yr[c] + res(c) - mean) * rstd * lg[c] + lb[c];
    for (int c = lane; c < g.d0; c += kWave) orow[O + c] = zs[r * g.d0 + c];
  }
  __syncthreads();
  // stage 4: h1 (buf1) = relu(cat . fc1^T + b1)
  for (int nb = wave; nb < (g.emb + 31) / 32; nb += kChainWaves)
    chain_block<true, true>(buf0, LD, 32, g.fc1_w, g.Kc, g.emb, nb * 32, O + g.d0, g.fc1_b, 1, buf1, LD, 0, lane);
  __syncthreads();
  // stage 5: out = h1 . fc2^T + b2  -> global
  for (int nb = wave; nb < (g.emb_out + 31) / 32; nb += kChainWaves)
    chain_block<true, false>(buf1, LD, rows, g.fc2_w, g.Ep, g.emb_out, nb * 32, g.emb, g.fc2_b, 0, g.out + m0 * g.ldo, g.ldo, 0, lane);
}

// ---------------------------------------------------------------------------
// The same tail as tgat_post_chain_kernel, TRANSPOSED and with the weights shared through LDS: tgat_chain64_kernel.
//
// Tiles of 16 rows and Y^T = W . X^T on v_mfma_f32_16x16x4_f32: the WEIGHT rows are the MFMA's A operand (lane (n, kq) holds
// W[n][16 kb + 4 kq + j]) and the activations its B operand (lane (r, kq) holds X[r][16 kb + 4 kq + j]).  The result lands as
// lane (r, rq) <- Y[r][16 nb + 4 rq + j]: four consecutive columns of the lane's own row, so a stage's output goes back to
// the tile's activation slab in LDS as 16-byte row pieces and the next stage reads it the same way; no stage but the last
// touches global memory.  A workgroup is 64 rows = 4 tiles x 2 waves (the two waves of a tile own the even and the odd
// 16-column output blocks: <= 6 independent accumulators per k-step each); 12 600 rows are 197 workgroups, one per CU.
//
// What the first versions taught (per-phase cycle stamps, 12 600 rows; the MFMAs alone are 51k cycles per SIMD):
// * one wave per tile streaming its own copy of the weights from L2: 66 us -- a CU takes ~12 bytes per clock from L2 however
//   the loads are shaped (row-major or tiled weights, deeper prefetch, MFMAs removed: all the same), and that was 1.8 MB;
// * so the waves of a CU share each weight chunk through LDS (24 tiles = 24 KB by LDS-DMA, 0.4 MB per CU in all), the weights
//   come pre-tiled (tgmx_tgat_tile16) in the order the chunks are consumed, and the five GEMMs are one chunk stream;
// * every dependent LDS access of a one-wave-per-SIMD kernel is exposed: biases / LayerNorm vectors are zero-padded float4
//   reads, LayerNorm runs on the accumulators (the row statistics are two shuffles and one LDS exchange away), padding
//   columns are written as zeros so that no activation read needs a guard;
// * a burst of VMEM instructions blocks the wave (~50 cycles per KB at the CU's address unit): memory instructions are woven
//   one at a time between groups of MFMAs, by hand, pinned with scheduling fences (C64Gemm::run).
// 159k cycles (tgat_post_chain_kernel, 69 us) -> 119k (51 us).  Still 2.3 x the MFMA time: the per-chunk bookkeeping between
// the MFMA groups does not overlap with them (measured: removing DMA, tile reads and barriers leaves ~1.9k cycles per chunk
// iteration); the next step is fewer, longer chunks without the k-step waste that a 48-tile chunk costs at these widths.
// ---------------------------------------------------------------------------
using floatx4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;

constexpr int kC64Tiles = 4;               // 16-row tiles per workgroup
constexpr int kC64Waves = 2 * kC64Tiles;   // two waves per tile (each owns every other output block): 2 waves per SIMD, so one
                                           // wave's memory-instruction issue and LDS latencies hide behind the other's MFMAs
constexpr int kC64Threads = kC64Waves * kWave;
constexpr int kC64ChunkTiles = 24;  // weight tiles (1 KB each) per LDS buffer
constexpr int kC64Ring = 3;          // LDS buffers: chunk p lives in buffer p % kC64Ring and is issued kC64Ring - 1 chunks ahead
constexpr int kC64MaxB = 12;        // output blocks per stage: N <= 192
__host__ __device__ constexpr int c64_steps(int nb) { return kC64ChunkTiles / nb < 8 ? kC64ChunkTiles / nb : 8; }  // k-steps per chunk

// The weights of the five GEMMs are ONE stream of chunks (<= 24 tiles each) through a ring of three LDS buffers: chunk p
// lives in buffer p % 3 and is issued as LDS-DMA two chunks before it is consumed, across GEMM boundaries, so a DMA has two
// chunks of MFMAs (~6000 cycles) to land and no GEMM starts with an exposed load.  Wave w moves tiles w, w + 8, ... of a
// chunk, 1 KB per instruction (destination = wave-uniform tile base + 16 bytes per lane), and ALWAYS issues kC64Issue
// instructions per chunk (a tile it has no use for goes to a dummy slot): the wait at the end of an iteration is the
// counted `s_waitcnt vmcnt(kC64Issue)` -- everything but the newest chunk has landed -- followed by a raw s_barrier
// (__syncthreads would drain the queue).
constexpr int kC64Issue = kC64ChunkTiles / kC64Waves;

struct C64Stream {
  // four kinds of GEMM: V = W_V (H of them, `vstride` floats apart), O = W_O, F1 = fc1, F2 = fc2; per kind: tiled image, tiles
  // per chunk, chunks, tiles.  (Plain scalars selected by compares, never indexed: the table stays in scalar registers.)
  const float *W_V, *W_O, *W_F1, *W_F2;
  int CT_V, CT_O, CT_F1, CT_F2, nch_V, nch_O, nch_F1, nch_F2, nt_V, nt_O, nt_F1, nt_F2;
  long long vstride;
  int H;
};

__device__ __forceinline__ void c64_geom(int NB, int K, int& CT, int& nch, int& nt) {
  const int KB = (K + 15) / 16, CK = c64_steps(NB);
  CT = CK * NB; nch = (KB + CK - 1) / CK; nt = KB * NB;
}

// The chunk under the cursor as kC64Issue (source, LDS destination) pairs for this wave -- a tile it has no use for (or
// anything past the end of the stream) is a dummy move -- and advance the cursor.  The moves themselves are issued one at a
// time between groups of MFMAs (c64_dma).
struct C64Moves {
  const float* src[kC64Issue];
  float* dst[kC64Issue];
};
__device__ __forceinline__ C64Moves c64_next_moves(const C64Stream st, int& ck, int& cc, int& cp, float* __restrict__ wbuf, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  const bool live = ck < st.H + 3;
  const int kind = ck < st.H ? 0 : ck - st.H + 1;
  const float* Wt = kind == 0 ? st.W_V + ck * st.vstride : kind == 1 ? st.W_O : kind == 2 ? st.W_F1 : st.W_F2;
  const int CT = kind == 0 ? st.CT_V : kind == 1 ? st.CT_O : kind == 2 ? st.CT_F1 : st.CT_F2;
  const int nch = kind == 0 ? st.nch_V : kind == 1 ? st.nch_O : kind == 2 ? st.nch_F1 : st.nch_F2;
  const int ntiles = kind == 0 ? st.nt_V : kind == 1 ? st.nt_O : kind == 2 ? st.nt_F1 : st.nt_F2;
  const int first = cc * CT, count = live ? CT : 0;
  float* buf = wbuf + (cp % kC64Ring) * (kC64ChunkTiles * 256);
  float* dummy = wbuf + kC64Ring * (kC64ChunkTiles * 256) + wave * 256;
  C64Moves m;
#pragma unroll
  for (int i = 0; i < kC64Issue; ++i) {
    const int t = i * kC64Waves + wave;
    const bool ok = t < count && first + t < ntiles;
    m.src[i] = Wt + (long long)(ok ? first + t : 0) * 256 + lane * 4;
    m.dst[i] = ok ? buf + t * 256 : dummy;
  }
  ++cp;
  const int c1 = cc + (live ? 1 : 0);
  const bool wrap = c1 == nch;
  cc = wrap ? 0 : c1;
  ck += wrap ? 1 : 0;
  return m;
}
// One LDS-DMA instruction, as inline assembly ON PURPOSE: hipcc tracks the builtin as an LDS store and puts a vmcnt(0) in front
// of the next ds_read it cannot prove disjoint -- which drains the chunk that is supposed to stay in flight.  Written this
// way the compiler knows nothing about it; the counted waits in c64_sync are what orders it (the compiler's own vmcnt waits
// for ordinary loads can only be too strict with unknown operations in the queue, never too weak).
__device__ __forceinline__ void c64_dma(const float* src, float* dst) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)dst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(lds)
               : "memory");
}

// all but the newest chunk have landed and every wave is done with the buffer the next issue overwrites
template <int INFLIGHT>
__device__ __forceinline__ void c64_sync() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(INFLIGHT) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ void c64_lds_barrier() {  // LDS writes of the workgroup visible; never drains the DMA queue
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// One GEMM of the chain for the eight waves of the workgroup at once (every thread must call it): wave (tile, HALF) computes
//   acc[i] (i < NBW) <- W[16 b .. 16 b + 16, 0 .. K) . X[the tile's 16 rows, 0 .. K)^T   for its blocks b = HALF + 2 i < NBT,
// the weights being the next chunks of the stream (q = position of this GEMM's first chunk, advanced).  XG: the activations
// come from global memory (the lane's zbar row, padded length Kxp) and xc holds the first chunk's on entry; xnext = the
// next GEMM's row, whose first chunk is fetched into xc during this one's last.  Otherwise xrow is the lane's row of the
// tile's LDS slab (written by both waves of the tile: the GEMM starts with a barrier).
template <int NBT, int HALF, bool XG>
struct C64Gemm {
  static constexpr int CK = c64_steps(NBT);          // k-steps per chunk
  static constexpr int CT = CK * NBT;                // tiles per chunk
  static constexpr int NBW = (NBT + 1 - HALF) / 2;   // this wave's blocks
  static constexpr int NBS = NBW > 0 ? NBW : 1;      // (array extents)

  // One activation vector (k-step kb) of the lane's row.  Slab rows need no guard: every stage writes its padding columns as
  // zeros (zero-padded tiles and biases), so a column past K is either zero or multiplies a zero weight.  zbar's padding
  // [C, Cp) is not initialised: the last k-block (only) is masked.
  // A k-step past KB (the tail of a GEMM's last chunk) returns zeros: its MFMAs then add 0 x whatever (finite) tiles the ring holds.
  static __device__ __forceinline__ float4 xload(const float* __restrict__ xrow, int kb, int K, int Kxp, int lane) {
    const int KB = (K + 15) / 16, k4l = (lane >> 4) * 4;
    const bool live = kb < KB;  // wave-uniform
    kb = live ? kb : KB - 1;
    const int k4 = kb * 16 + k4l;
    float4 x;
    if constexpr (!XG) {
      x = *reinterpret_cast<const float4*>(xrow + k4);
    } else {
      const int kx_last = Kxp - 4;
      x = *reinterpret_cast<const float4*>(xrow + (k4 < kx_last ? k4 : kx_last));
      if (kb == KB - 1) x = float4{k4 < K ? x.x : 0.f, k4 + 1 < K ? x.y : 0.f, k4 + 2 < K ? x.z : 0.f, k4 + 3 < K ? x.w : 0.f};
    }
    return live ? x : float4{0.f, 0.f, 0.f, 0.f};
  }

  // The chunk loop is software-pipelined BY HAND and pinned with scheduling fences, because what stalls this kernel is issue
  // order: a VMEM instruction blocks the wave while the CU's address unit takes it (~50 cycles per KB, and the waves come
  // out of a barrier together: the DMAs of a chunk issued in a burst = 1200 cycles of idle matrix pipe per chunk, 28k of
  // the first version's 130k cycles), and a tile read issued right before its MFMAs exposes the LDS latency 60 times.  So,
  // between every two groups of NBW MFMAs: at most one VMEM instruction (this wave's three DMAs of chunk q + 2 in groups
  // CK .. CK + 2; after the last group of every k-step the next chunk's activation vector for that step, straight into the
  // register the step just released) and a quarter of the next k-step's tile reads.  The wait + barrier that retires chunk
  // q + 1 sits before the LAST group of MFMAs of chunk q, followed by the first tile reads of chunk q + 1, which that last
  // group covers.  (Running the two waves that share a SIMD half a group apart -- one sleeping 128 cycles after every barrier --
  // was measured: 59.9 vs 58.7 us without.)
  static __device__ __forceinline__ void run(const C64Stream st, int& ck, int& cc, int& cp, int& q, int K, const float* __restrict__ xrow, int Kxp,
                                             float4 (&xc)[CK], const float* __restrict__ xnext, float* __restrict__ wbuf, floatx4 (&acc)[NBS], int tid) {
    static_assert(4 * CK > CK + kC64Issue, "not enough MFMA groups per chunk to carry the loads");
    constexpr int Q4 = (NBW + 3) / 4;  // tile reads per MFMA group
    const int lane = tid & 63;
    const int KB = (K + 15) / 16, nchunks = (KB + CK - 1) / CK;
#pragma unroll
    for (int i = 0; i < NBS; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!XG) {
      c64_lds_barrier();  // the previous stage's slab writes of both waves of the tile
#pragma unroll
      for (int s = 0; s < CK; ++s) xc[s] = xload(xrow, s, K, Kxp, lane);
    }
    float4 w[2][NBS];
    {
      const float* wl = wbuf + (q % kC64Ring) * (kC64ChunkTiles * 256) + lane * 4;
#pragma unroll
      for (int i = 0; i < NBW; ++i) w[0][i] = *reinterpret_cast<const float4*>(wl + (HALF + 2 * i) * 256);
    }
    C64Moves mv = c64_next_moves(st, ck, cc, cp, wbuf, tid);  // position q + 2
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks;
      const float* xsrc = more ? xrow : (XG && xnext ? xnext : xrow);
      const int cnext = more ? c + 1 : 0;
      C64Moves mv_next = mv;
      const float* wl = wbuf + (q % kC64Ring) * (kC64ChunkTiles * 256) + lane * 4;
      const float* wl_next = wbuf + ((q + 1) % kC64Ring) * (kC64ChunkTiles * 256) + lane * 4;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < CK; ++s) {
        float4(&wc)[NBS] = w[s & 1];  // k-step s of a chunk reads register set s & 1 (an odd CK copies once per chunk)
        float4(&wn)[NBS] = w[(s & 1) ^ 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int grp = 4 * s + j;
          if (s == CK - 1 && j == 3) {
            // chunk q + 1 has landed and everybody is done reading chunk q.  In flight may stay: the newest kC64Issue VMEM
            // operations, all of which are younger than chunk q + 1's DMAs
            c64_sync<(kC64Ring - 2) * kC64Issue>();
            if (more) {
#pragma unroll
              for (int i = 0; i < NBW; ++i) wn[i] = *reinterpret_cast<const float4*>(wl_next + (HALF + 2 * i) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int i = 0; i < NBW; ++i) {
            const float a = j == 0 ? wc[i].x : j == 1 ? wc[i].y : j == 2 ? wc[i].z : wc[i].w;
            const float b = j == 0 ? xc[s].x : j == 1 ? xc[s].y : j == 2 ? xc[s].z : xc[s].w;
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          // the next iteration's address arithmetic in the shadow of these MFMAs
          if (grp == CK + kC64Issue && more) mv_next = c64_next_moves(st, ck, cc, cp, wbuf, tid);
          if (grp >= CK && grp < CK + kC64Issue) c64_dma(mv.src[grp - CK], mv.dst[grp - CK]);
          if (j == 3) xc[s] = xload(xsrc, cnext * CK + s, K, Kxp, lane);  // step s is done with xc[s]
          if (s + 1 < CK) {
#pragma unroll
            for (int i = j * Q4; i < (j + 1) * Q4 && i < NBW; ++i) wn[i] = *reinterpret_cast<const float4*>(wl + ((s + 1) * NBT + HALF + 2 * i) * 256);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr ((CK & 1) == 1) {  // the next chunk's step 0 was read into w[1]: step 0 always reads w[0]
#pragma unroll
        for (int i = 0; i < NBW; ++i) w[0][i] = w[1][i];
      }
      ++q;
      mv = mv_next;
    }
  }

  // run() for stage 1 with BOTH heads' output blocks in one pass (NBT = 2 x blocks per head; block b belongs to head b / (NBT / 2)): the
  // weights are ONE stream of chunks over the stacked, per-head zero-padded W_V (tgmx_tgat_layer_t.W_V_t16c), the two heads' zbar rows come
  // from global memory (row0, row1; the first chunk of both in xc0 / xc1 on entry).  Per k-step a wave issues 4 NBW MFMAs in groups of NBW
  // -- twice the per-head pass's -- against the same per-group bookkeeping, and the stage is half as many chunks.
  static __device__ __forceinline__ void run_heads(const C64Stream st, int& ck, int& cc, int& cp, int& q, int K, const float* __restrict__ row0,
                                                   const float* __restrict__ row1, int Kxp, float4 (&xc0)[CK], float4 (&xc1)[CK],
                                                   float* __restrict__ wbuf, floatx4 (&acc)[NBS], int tid) {
    static_assert(XG && NBT % 2 == 0, "two heads, activations from global memory");
    static_assert(4 * CK > CK + kC64Issue, "not enough MFMA groups per chunk to carry the loads");
    constexpr int Q4 = (NBW + 3) / 4, HBc = NBT / 2;
    const int lane = tid & 63;
    const int KB = (K + 15) / 16, nchunks = (KB + CK - 1) / CK;
#pragma unroll
    for (int i = 0; i < NBS; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float4 w[2][NBS];
    {
      const float* wl = wbuf + (q % kC64Ring) * (kC64ChunkTiles * 256) + lane * 4;
#pragma unroll
      for (int i = 0; i < NBW; ++i) w[0][i] = *reinterpret_cast<const float4*>(wl + (HALF + 2 * i) * 256);
    }
    C64Moves mv = c64_next_moves(st, ck, cc, cp, wbuf, tid);  // position q + 2
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks;
      const int cnext = more ? c + 1 : c;  // (the last chunk refills itself: harmless, keeps the body branch-free)
      C64Moves mv_next = mv;
      const float* wl = wbuf + (q % kC64Ring) * (kC64ChunkTiles * 256) + lane * 4;
      const float* wl_next = wbuf + ((q + 1) % kC64Ring) * (kC64ChunkTiles * 256) + lane * 4;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < CK; ++s2) {
        float4(&wc)[NBS] = w[s2 & 1];
        float4(&wn)[NBS] = w[(s2 & 1) ^ 1];
        float4 x0n = xc0[s2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int grp = 4 * s2 + j;
          if (s2 == CK - 1 && j == 3) {
            c64_sync<(kC64Ring - 2) * kC64Issue>();
            if (more) {
#pragma unroll
              for (int i = 0; i < NBW; ++i) wn[i] = *reinterpret_cast<const float4*>(wl_next + (HALF + 2 * i) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int i = 0; i < NBW; ++i) {
            const bool h1 = (HALF + 2 * i) >= HBc;  // compile-time per i
            const float a = j == 0 ? wc[i].x : j == 1 ? wc[i].y : j == 2 ? wc[i].z : wc[i].w;
            const float b = h1 ? (j == 0 ? xc1[s2].x : j == 1 ? xc1[s2].y : j == 2 ? xc1[s2].z : xc1[s2].w)
                               : (j == 0 ? xc0[s2].x : j == 1 ? xc0[s2].y : j == 2 ? xc0[s2].z : xc0[s2].w);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (grp == CK + kC64Issue && more) mv_next = c64_next_moves(st, ck, cc, cp, wbuf, tid);
          if (grp >= CK && grp < CK + kC64Issue) c64_dma(mv.src[grp - CK], mv.dst[grp - CK]);
          if (j == 2) x0n = xload(row0, cnext * CK + s2, K, Kxp, lane);  // head 0's vector of the next chunk: parked until step s2 is done with xc0[s2]
          if (j == 3) {
            xc1[s2] = xload(row1, cnext * CK + s2, K, Kxp, lane);
            xc0[s2] = x0n;
          }
          if (s2 + 1 < CK) {
#pragma unroll
            for (int i = j * Q4; i < (j + 1) * Q4 && i < NBW; ++i) wn[i] = *reinterpret_cast<const float4*>(wl + ((s2 + 1) * NBT + HALF + 2 * i) * 256);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr ((CK & 1) == 1) {
#pragma unroll
        for (int i = 0; i < NBW; ++i) w[0][i] = w[1][i];
      }
      ++q;
      mv = mv_next;
    }
  }
};

// run f(integral_constant<NB>) for the (workgroup-uniform) block count nb in [1, MAXB]
template <int NB, int MAXB, class F>
__device__ __forceinline__ void chain64_dispatch(int nb, F&& f) {
  if (nb == NB) {
    f(std::integral_constant<int, NB>{});
  } else if constexpr (NB < MAXB) {
    chain64_dispatch<NB + 1, MAXB>(nb, f);
  }
}

__global__ __launch_bounds__(kC64Threads) void tgat_chain64_kernel(const ChainArgs g) {
  extern __shared__ __attribute__((aligned(16))) float c64_lds[];
  constexpr int NT = kC64Threads;
  const int O = g.O, H = g.H, dh = O / H, LD = g.LD, T = g.T, d = g.d, d0 = g.d0;
  const int nbO = (O + 15) / 16, nbE = (g.emb + 15) / 16, nbEo = (g.emb_out + 15) / 16;
  float* wbuf = c64_lds;                                             // [3][24 tiles x 256] weight chunks + one dummy tile per wave
  float* slabs = wbuf + kC64Ring * kC64ChunkTiles * 256 + kC64Waves * 256;  // [4 tiles][16, LD] activations
  float* br = slabs + kC64Tiles * 16 * LD;          // [16 nbO]  b_O + the residual's time part cos(tb), zero-padded
  float* lg = br + 16 * nbO;                        // [16 nbO]  LayerNorm weight, zero-padded
  float* lb = lg + 16 * nbO;                        // [16 nbO]  LayerNorm bias, zero-padded
  float* b1 = lb + 16 * nbO;                        // [16 nbE]  zero-padded
  float* b2 = b1 + 16 * nbE;                        // [16 nbEo] zero-padded
  float* stat = b2 + 16 * nbEo;                     // [4 tiles][2 halves][16 rows] LayerNorm partial sums
  float* xs = stat + kC64Waves * 16;                // [64, d]  residual feature part
  float* zs = xs + 64 * d;                          // [64, d0] skip features
  // the two waves of a tile own the even / the odd output blocks (ceil vs floor of half the blocks: 912 vs 796 MFMAs a wave at the headline
  // widths).  Waves w and w + 4 share a SIMD: tiles 2 and 3 swap the roles, so every SIMD carries one heavier and one lighter wave
  // (59.2 -> 57.9 us)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = wave >> 1, half = (wave ^ (wave >> 2)) & 1;
  const int r = lane & 15, rq = lane >> 4;
  const long long w0 = (long long)blockIdx.x * 64;  // first row of the workgroup
  const long long m0 = w0 + tile * 16;              // first row of this wave's tile
  const int wrows = (g.R - w0) < 64 ? (int)(g.R - w0) : 64;
  if (!rows_live(g.segs, w0, w0 + wrows)) return;  // compact rows: nothing of this workgroup's 64 rows exists
  const int rows = g.R - m0 < 0 ? 0 : (g.R - m0 < 16 ? (int)(g.R - m0) : 16);  // 0: the wave only helps with the copies
  float* slab = slabs + tile * 16 * LD;
  const int HB = (dh + 15) / 16, KBc = (g.C + 15) / 16;
  int q = 0;
  C64Stream st;
  int ck = 0, cc = 0, cp = 0;
  const bool both = g.W_Vc != nullptr;  // stage 1 as one pass over both heads
  st.H = both ? 1 : H; st.vstride = (long long)HB * KBc * 256;
  st.W_V = both ? g.W_Vc : g.W_V; st.W_O = g.W_O; st.W_F1 = g.fc1_w; st.W_F2 = g.fc2_w;
  c64_geom(both ? 2 * HB : HB, g.C, st.CT_V, st.nch_V, st.nt_V);
  c64_geom(nbO, O, st.CT_O, st.nch_O, st.nt_O);
  c64_geom(nbE, O + d0, st.CT_F1, st.nch_F1, st.nt_F1);
  c64_geom(nbEo, g.emb, st.CT_F2, st.nch_F2, st.nt_F2);
  // the slabs and the weight ring start as zeros: a partial chunk's trailing k-steps multiply whatever the ring holds with zeros
  for (int e = tid * 4; e < kC64Ring * kC64ChunkTiles * 256 + kC64Waves * 256 + kC64Tiles * 16 * LD; e += NT * 4)
    *reinterpret_cast<float4*>(wbuf + e) = float4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int u = 0; u < kC64Ring - 1; ++u) {  // the first weight chunks fly during the rest of the prologue
    const C64Moves mv = c64_next_moves(st, ck, cc, cp, wbuf, tid);
#pragma unroll
    for (int i = 0; i < kC64Issue; ++i) c64_dma(mv.src[i], mv.dst[i]);
  }
  {
    // row constants and this workgroup's residual / skip inputs: every global load first, then the LDS stores
    const int t0 = O - T;
    const int c = tid;  // nbO, nbE, nbEo <= 12: one element per thread covers every padded vector
    const float v_bo = c < O ? g.b_O[c] : 0.f, v_tb = (c >= t0 && c < O) ? g.tb[c - t0] : 0.f;
    const float v_lg = c < O ? g.ln_g[c] : 0.f, v_lb = c < O ? g.ln_b[c] : 0.f;
    const float v_b1 = c < g.emb ? g.fc1_b[c] : 0.f, v_b2 = c < g.emb_out ? g.fc2_b[c] : 0.f;
    const float v_x = c < wrows * d ? g.x[(w0 + c / d) * g.ldx + (c % d)] : 0.f;
    const float v_z = c < wrows * d0 ? g.z0[w0 * d0 + c] : 0.f;
    if (c < 16 * nbO) {
      br[c] = v_bo + ((c >= t0 && c < O) ? cos_t2v(v_tb) : 0.f);
      lg[c] = v_lg;
      lb[c] = v_lb;
    }
    if (c < 16 * nbE) b1[c] = v_b1;
    if (c < 16 * nbEo) b2[c] = v_b2;
    if (c < 64 * d) xs[c] = v_x;
    if (c < 64 * d0) zs[c] = v_z;
    for (int e = tid + NT; e < wrows * d; e += NT) xs[e] = g.x[(w0 + e / d) * g.ldx + (e % d)];
    for (int e = tid + NT; e < wrows * d0; e += NT) zs[e] = g.z0[w0 * d0 + e];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler does not know about the DMAs above
  __syncthreads();  // both chunks have landed: the stream's first wait finds nothing in flight, which a counted wait allows

  // every stage below runs as <blocks of the stage, this wave's half>: both are workgroup- resp. wave-uniform
  auto with_half = [&](auto nbc, auto&& f) __attribute__((always_inline)) {
    if (half == 0) f(nbc, std::integral_constant<int, 0>{});
    else f(nbc, std::integral_constant<int, 1>{});
  };

  // stage 1: oattn (slab) -- per head, the lane's zbar row straight from global.  Head h writes columns h dh .. h dh + 16 HB:
  // its zero padding (the tiles are zero past dh) is overwritten by head h + 1 (a barrier later: the chunk barriers of head
  // h + 1's GEMM order the two waves' writes), the last head's lands past O where no stage looks
  {
    long long grow = m0 + (r < rows ? r : rows - 1);
    grow = grow < g.R ? (grow < 0 ? 0 : grow) : g.R - 1;
    const float* zrow = g.zbar + grow * g.ld_zbar;
    if (both) {
      chain64_dispatch<1, kC64MaxB>(2 * HB, [&](auto nbc) __attribute__((always_inline)) {
        with_half(nbc, [&](auto nbc2, auto hc) __attribute__((always_inline)) {
          constexpr int NBT = decltype(nbc2)::value, HALF = decltype(hc)::value;
          if constexpr (NBT % 2 == 0) {
            using G = C64Gemm<NBT, HALF, true>;
            float4 xc0[G::CK], xc1[G::CK];
#pragma unroll
            for (int s = 0; s < G::CK; ++s) {
              xc0[s] = G::xload(zrow, s, g.C, g.Cp, lane);
              xc1[s] = G::xload(zrow + g.Cp, s, g.C, g.Cp, lane);
            }
            floatx4 acc[G::NBS];
            G::run_heads(st, ck, cc, cp, q, g.C, zrow, zrow + g.Cp, g.Cp, xc0, xc1, wbuf, acc, tid);
            // head 0 writes its dh real columns only: its zero padding would land on head 1's first columns, written in the same pass
#pragma unroll
            for (int i = 0; i < G::NBW; ++i) {
              constexpr int HBc = NBT / 2;
              const int b = HALF + 2 * i, h = b >= HBc ? 1 : 0, nb = b - h * HBc;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int cl = 16 * nb + 4 * rq + j;
                if (h == 1 || cl < dh) slab[r * LD + h * dh + cl] = acc[i][j];
              }
            }
          }
        });
      });
    } else
    chain64_dispatch<1, kC64MaxB>(HB, [&](auto nbc) __attribute__((always_inline)) {
      with_half(nbc, [&](auto nbc2, auto hc) __attribute__((always_inline)) {
        constexpr int NBT = decltype(nbc2)::value, HALF = decltype(hc)::value;
        using G = C64Gemm<NBT, HALF, true>;
        float4 xc[G::CK];
#pragma unroll
        for (int s = 0; s < G::CK; ++s) xc[s] = G::xload(zrow, s, g.C, g.Cp, lane);
        for (int h = 0; h < H; ++h) {
          floatx4 acc[G::NBS];
          const bool last = h + 1 == H;
          G::run(st, ck, cc, cp, q, g.C, zrow + (long long)h * g.Cp, g.Cp, xc, last ? nullptr : zrow + (long long)(h + 1) * g.Cp, wbuf, acc, tid);
#pragma unroll
          for (int i = 0; i < G::NBW; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) slab[r * LD + h * dh + 16 * (HALF + 2 * i) + 4 * rq + j] = acc[i][j];
          }
        }
      });
    });
  }
  // stage 2 + 3: cat = [LayerNorm(oattn . W_O^T + b_O + residual) | z0].  The two waves of a tile hold whole rows between them
  // (4 lanes x 4 NBW registers each): the statistics are two shuffles and one LDS exchange away, y never goes to LDS
  chain64_dispatch<1, kC64MaxB>(nbO, [&](auto nbc) __attribute__((always_inline)) {
    with_half(nbc, [&](auto nbc2, auto hc) __attribute__((always_inline)) {
      constexpr int NBT = decltype(nbc2)::value, HALF = decltype(hc)::value;
      using G = C64Gemm<NBT, HALF, false>;
      float4 xc[G::CK];
      floatx4 acc[G::NBS];
      G::run(st, ck, cc, cp, q, O, slab + r * LD, 0, xc, nullptr, wbuf, acc, tid);
      float y[G::NBS][4];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < G::NBW; ++i) {
        const int c0 = 16 * (HALF + 2 * i) + 4 * rq;
        const float4 b = *reinterpret_cast<const float4*>(br + c0);
        y[i][0] = acc[i][0] + b.x; y[i][1] = acc[i][1] + b.y; y[i][2] = acc[i][2] + b.z; y[i][3] = acc[i][3] + b.w;
        if (16 * (HALF + 2 * i) < d) {  // the residual's feature part: the first ceil(d / 16) blocks only
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (c0 + j < d) y[i][j] += xs[(tile * 16 + r) * d + c0 + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += c0 + j < O ? y[i][j] : 0.f;
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      float* mystat = stat + (tile * 2 + HALF) * 16;
      const float* other = stat + (tile * 2 + (HALF ^ 1)) * 16;
      if (rq == 0) mystat[r] = sum;
      c64_lds_barrier();
      const float mean = (sum + other[r]) / (float)O;
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < G::NBW; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = 16 * (HALF + 2 * i) + 4 * rq + j < O ? y[i][j] - mean : 0.f;
          var += u * u;
        }
      }
      var += __shfl_xor(var, 16);
      var += __shfl_xor(var, 32);
      c64_lds_barrier();  // everybody has read the sums
      if (rq == 0) mystat[r] = var;
      c64_lds_barrier();
      const float rstd = 1.0f / sqrtf((var + other[r]) / (float)O + g.eps);
#pragma unroll
      for (int i = 0; i < G::NBW; ++i) {
        const int c0 = 16 * (HALF + 2 * i) + 4 * rq;
        const float4 gw = *reinterpret_cast<const float4*>(lg + c0), gb = *reinterpret_cast<const float4*>(lb + c0);
        // columns past O: weight and bias are zero-padded, so they are written as zeros
        *reinterpret_cast<float4*>(slab + r * LD + c0) = float4{(y[i][0] - mean) * rstd * gw.x + gb.x, (y[i][1] - mean) * rstd * gw.y + gb.y,
                                                                 (y[i][2] - mean) * rstd * gw.z + gb.z, (y[i][3] - mean) * rstd * gw.w + gb.w};
      }
      if (d0 > 0) {
        c64_lds_barrier();  // the skip columns may share a block with LayerNorm's zero padding written by the other wave
        if (HALF == 0) {
          for (int e = lane; e < 16 * d0; e += 64) {
            const int row = e / d0, c = e - row * d0;
            slab[row * LD + O + c] = row < rows ? zs[(tile * 16 + row) * d0 + c] : 0.f;
          }
        }
      }
    });
  });
  // stage 4: h1 = relu(cat . fc1^T + b1), in place
  chain64_dispatch<1, kC64MaxB>(nbE, [&](auto nbc) __attribute__((always_inline)) {
    with_half(nbc, [&](auto nbc2, auto hc) __attribute__((always_inline)) {
      constexpr int NBT = decltype(nbc2)::value, HALF = decltype(hc)::value;
      using G = C64Gemm<NBT, HALF, false>;
      float4 xc[G::CK];
      floatx4 acc[G::NBS];
      G::run(st, ck, cc, cp, q, O + d0, slab + r * LD, 0, xc, nullptr, wbuf, acc, tid);
#pragma unroll
      for (int i = 0; i < G::NBW; ++i) {
        const int c0 = 16 * (HALF + 2 * i) + 4 * rq;
        const float4 b = *reinterpret_cast<const float4*>(b1 + c0);
        float4 v{acc[i][0] + b.x, acc[i][1] + b.y, acc[i][2] + b.z, acc[i][3] + b.w};  // columns past emb: zero tiles + zero bias
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        *reinterpret_cast<float4*>(slab + r * LD + c0) = v;
      }
    });
  });
  // stage 5: out = h1 . fc2^T + b2 -> global
  chain64_dispatch<1, kC64MaxB>(nbEo, [&](auto nbc) __attribute__((always_inline)) {
    with_half(nbc, [&](auto nbc2, auto hc) __attribute__((always_inline)) {
      constexpr int NBT = decltype(nbc2)::value, HALF = decltype(hc)::value;
      using G = C64Gemm<NBT, HALF, false>;
      float4 xc[G::CK];
      floatx4 acc[G::NBS];
      G::run(st, ck, cc, cp, q, g.emb, slab + r * LD, 0, xc, nullptr, wbuf, acc, tid);
      if (r < rows) {
        float* orow = g.out + (m0 + r) * g.ldo;
        const bool vec = (g.ldo & 3) == 0;
#pragma unroll
        for (int i = 0; i < G::NBW; ++i) {
          const int c0 = 16 * (HALF + 2 * i) + 4 * rq;
          const float4 b = *reinterpret_cast<const float4*>(b2 + c0);
          const float4 v{acc[i][0] + b.x, acc[i][1] + b.y, acc[i][2] + b.z, acc[i][3] + b.w};
          if (vec && c0 + 3 < g.emb_out) {
            *reinterpret_cast<float4*>(orow + c0) = v;
          } else {
            if (c0 < g.emb_out) orow[c0] = v.x;
            if (c0 + 1 < g.emb_out) orow[c0 + 1] = v.y;
            if (c0 + 2 < g.emb_out) orow[c0 + 2] = v.z;
            if (c0 + 3 < g.emb_out) orow[c0 + 3] = v.w;
          }
        }
      }
    });
  });
}

// ---------------------------------------------------------------------------
// Per-row attention over the k sampled neighbor slots (one wave per row).
// ---------------------------------------------------------------------------
struct AttnArgs {
  const float* qf;    // [R, H, C] folded queries, C = d + D + T laid out [nbr | edge | time]
  const float* nbrf;  // [R, k, d] neighbor node features / embeddings
  const float* ex;    // [R, k, D] sampled edge features
  const int64_t* seed_t;  // [R]
  const int64_t* nbr_t;   // [R, k]
  const int32_t* nbr_id;  // [R, k] (-1 = empty slot -> masked)
  const float* tw;    // [T] Time2Vec weight
  const float* tb;    // [T] Time2Vec bias
  const float* tfeat;       // optional [R, k, T]: precomputed neighbor time features (then tw/tb/times unused)
  const unsigned char* mask;  // optional [R, k]: valid-neighbor mask (then nbr_id unused)
  float* zbar;        // [R, H, C]
  long long R;
  int d, D, T, k, C;
  float scale;        // dh^-1/2
  int Cs;             // stride (floats) between consecutive heads of qf / zbar rows (>= C; padded layouts)
  float* probs;       // optional [R, H, k]: the attention weights (BEFORE dropout), saved for the backward pass
  DropoutArgs drop;   // training: dropout on the attention weights (attention.py:119), element index ((row0 + r) * H + h) * k + s
  long long drop_row0;
  // Several levels of the hop tree in ONE launch (they share the layer's weights; qf / zbar / probs are contiguous
  // across levels).  Rows [seg_begin[i], seg_begin[i + 1]) read level i's sampler outputs; the pointers are biased by
  // the host so that the GLOBAL row index addresses them (ex_i - seg_begin[i] * k * D, ...).  n_seg <= 1: the plain
  // fields above.
  int n_seg;
  long long seg_begin[TGMX_TGAT_MAX_LAYERS];
  const float* seg_nbrf[TGMX_TGAT_MAX_LAYERS];
  const float* seg_ex[TGMX_TGAT_MAX_LAYERS];
  // edge features by id (tgmx_tgat_hop_t.nbr_eid): slot (r, s) reads seg_table[i][eid] instead of seg_ex[i][r, s] (register kernel only)
  const int32_t* seg_eid[TGMX_TGAT_MAX_LAYERS];
  const float* seg_table[TGMX_TGAT_MAX_LAYERS];
  const int64_t* seg_seed_t[TGMX_TGAT_MAX_LAYERS];
  const int64_t* seg_nbr_t[TGMX_TGAT_MAX_LAYERS];
  const int32_t* seg_nbr_id[TGMX_TGAT_MAX_LAYERS];
  // Compact rows (register kernel only; all NULL otherwise).  Row r of segment i stands for the ORIGINAL row seg_uniq[i][r] of its
  // level (biased by -seg_begin[i] like everything else; NULL: r itself): the sampler's outputs are read there, qf / zbar / probs at r.
  // Only the first *seg_live[i] rows of the segment exist (NULL: all of them).
  const int32_t* seg_uniq[TGMX_TGAT_MAX_LAYERS];
  const int32_t* seg_live[TGMX_TGAT_MAX_LAYERS];
  // Neighbor features by index: slot (ro, s) reads row seg_nidx[i][ro * k + s] of seg_ntab[i] (d floats per row; a negative index reads
  // row seg_npad[i]) instead of row ro * k + s of the dense block -- the compact rows of the level below, or node_x by node id.
  const int32_t* seg_nidx[TGMX_TGAT_MAX_LAYERS];
  const float* seg_ntab[TGMX_TGAT_MAX_LAYERS];
  long long seg_npad[TGMX_TGAT_MAX_LAYERS];
  int full_span;              // != 0: the register kernel runs every row through its all-slots body (A/B knob TGMX_ATTN_SPAN=0)
  // register kernel, a layer whose input is ONE scalar per row (d == 1, inference): qf is NOT a buffer -- qf[r, c] = qv[c] + qx[r] * qU[c],
  // evaluated by the lane that needs column c from the lane-ordered table tgmx_tgat_layer_t.qf_lane (the same fma as
  // tgat_qfold_small_kernel: bit-identical).  qf is ignored when qlane is set.
  const float* qlane;         // [H, 64, 16]
  const float* qx;            // [R] the rows' own feature
};

// the level (segment) of row r: wave-uniform, so this is scalar code
struct AttnLevel {
  const int32_t* eid = nullptr;
  const float* table = nullptr;
  const int32_t* uniq = nullptr;
  const int32_t* live = nullptr;
  const int32_t* nidx = nullptr;
  const float* ntab = nullptr;
  long long npad = 0, begin = 0;
  const float* nbrf;
  const float* ex;
  const int64_t* seed_t;
  const int64_t* nbr_t;
  const int32_t* nbr_id;
};
__device__ __forceinline__ AttnLevel attn_level(const AttnArgs& a, long long r) {
  AttnLevel v;
  v.nbrf = a.nbrf; v.ex = a.ex; v.seed_t = a.seed_t; v.nbr_t = a.nbr_t; v.nbr_id = a.nbr_id;
  v.eid = a.seg_eid[0]; v.table = a.seg_table[0];
  v.uniq = a.seg_uniq[0]; v.live = a.seg_live[0]; v.nidx = a.seg_nidx[0]; v.ntab = a.seg_ntab[0]; v.npad = a.seg_npad[0];
  v.begin = a.n_seg > 0 ? a.seg_begin[0] : 0;
  if (a.n_seg > 1) {
    int si = 0;
#pragma unroll
    for (int i = 1; i < TGMX_TGAT_MAX_LAYERS; ++i)
      if (i < a.n_seg && r >= a.seg_begin[i]) si = i;
    v.nbrf = a.seg_nbrf[si]; v.ex = a.seg_ex[si]; v.seed_t = a.seg_seed_t[si]; v.nbr_t = a.seg_nbr_t[si]; v.nbr_id = a.seg_nbr_id[si];
    v.eid = a.seg_eid[si]; v.table = a.seg_table[si];
    v.uniq = a.seg_uniq[si]; v.live = a.seg_live[si]; v.nidx = a.seg_nidx[si]; v.ntab = a.seg_ntab[si]; v.npad = a.seg_npad[si];
    v.begin = a.seg_begin[si];
  }
  return v;
}

// The value of lane (lane ^ O), O a power of two, WITHOUT the LDS crossbar: __shfl_xor is a ds_bpermute_b32 (an LDS-pipeline round trip,
// ~150 cycles), and a row of this kernel runs ~25 of them in dependent chains -- the score reduction's last four levels, the softmax's two
// butterflies -- with two waves per SIMD to hide them behind.  Data parallel primitives instead (a few cycles each): quad permutes for
// 1 and 2, half-row mirror + quad reverse for 4 (i ^ 7 ^ 3), a row rotation by 8, and the gfx950 row-pair / half-wave swaps for 16 and 32.
// Pure data movement: every sum and maximum keeps its operands and their order.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int O>
__device__ __forceinline__ float lane_xor(float v) {
  static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "a power of two below the wave size");
  if constexpr (O == 1) return dpp_mov<0xB1>(v);                      // quad_perm [1, 0, 3, 2]
  else if constexpr (O == 2) return dpp_mov<0x4E>(v);                 // quad_perm [2, 3, 0, 1]
  else if constexpr (O == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // row_half_mirror (i ^ 7), then quad_perm [3, 2, 1, 0] (i ^ 3)
  else if constexpr (O == 8) return dpp_mov<0x128>(v);                // row_ror:8
  else if constexpr (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane_id() & 16) ? r[0] : r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane_id() & 32) ? r[0] : r[1]);
  }
}
// max / sum over the lanes that differ in the bits >= LO (LO = H: the slots of one head), as the xor butterfly LO, 2 LO, ..., 32
template <int LO>
__device__ __forceinline__ float butterfly_max(float v) {
  if constexpr (LO <= 1) v = fmaxf(v, lane_xor<1>(v));
  if constexpr (LO <= 2) v = fmaxf(v, lane_xor<2>(v));
  if constexpr (LO <= 4) v = fmaxf(v, lane_xor<4>(v));
  if constexpr (LO <= 8) v = fmaxf(v, lane_xor<8>(v));
  if constexpr (LO <= 16) v = fmaxf(v, lane_xor<16>(v));
  return fmaxf(v, lane_xor<32>(v));
}
template <int LO>
__device__ __forceinline__ float butterfly_sum(float v) {
  if constexpr (LO <= 1) v += lane_xor<1>(v);
  if constexpr (LO <= 2) v += lane_xor<2>(v);
  if constexpr (LO <= 4) v += lane_xor<4>(v);
  if constexpr (LO <= 8) v += lane_xor<8>(v);
  if constexpr (LO <= 16) v += lane_xor<16>(v);
  return v + lane_xor<32>(v);
}

// sum over the 64 lanes of P[j], delivered to lane j: 63 shuffles instead of 64 * 6.
// (template steps: every register index must be a compile-time constant, or P spills to scratch)
template <int HALF>
__device__ __forceinline__ void reduce_scatter_step(float (&P)[64], int lane) {
  const bool upper = (lane & HALF) != 0;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const float send = upper ? P[i] : P[i + HALF];
    const float recv = __shfl_xor(send, HALF);
    const float keep = upper ? P[i + HALF] : P[i];
    P[i] = keep + recv;
  }
}
// the two widest steps as gfx950 lane swaps: v_permlane32_swap exchanges the upper half of one register with the lower
// half of another (v_permlane16_swap: odd 16-lane rows with even ones), so a' + b' IS keep + recv for both halves --
// two instructions per pair instead of two selects, a ds_bpermute and an add (same operands: bit-identical sums)
__device__ __forceinline__ void reduce_scatter_swap32(float (&P)[64]) {
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[i]), __float_as_uint(P[i + 32]), false, false);
    P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}
__device__ __forceinline__ void reduce_scatter_swap16(float (&P)[64]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[i]), __float_as_uint(P[i + 16]), false, false);
    P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}
__device__ __forceinline__ float reduce_scatter64(float (&P)[64], int lane) {
  reduce_scatter_swap32(P);
  reduce_scatter_swap16(P);
  reduce_scatter_step<8>(P, lane);
  reduce_scatter_step<4>(P, lane);
  reduce_scatter_step<2>(P, lane);
  reduce_scatter_step<1>(P, lane);
  return P[0];
}

// The same sums for N <= 64 live entries (N a power of two, P[0 .. N)): the steps whose stride is >= N cannot scatter (there is nothing
// left to split), so both partners add and keep all N entries; from stride N / 2 down it is the reduce-scatter above.  Every entry
// is summed over the 64 lanes in the order xor 32, 16, 8, 4, 2, 1 -- a + b == b + a, so WHICH lane accumulates does not matter and the
// sums are bit-identical to reduce_scatter64's.  Afterwards lane L holds entry L mod N.
template <int HALF, int N>
__device__ __forceinline__ void all_reduce_step(float (&P)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if constexpr (HALF == 32) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[i]), __float_as_uint(P[i]), false, false);
      P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (HALF == 16) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[i]), __float_as_uint(P[i]), false, false);
      P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
      P[i] = P[i] + lane_xor<HALF>(P[i]);
    }
  }
}
template <int HALF, int N>
__device__ __forceinline__ void scatter_step_n(float (&P)[N], int lane) {
  static_assert(2 * HALF <= N, "a scatter step halves 2 * HALF live entries");
  if constexpr (HALF == 32) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[i]), __float_as_uint(P[i + 32]), false, false);
      P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
  } else if constexpr (HALF == 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[i]), __float_as_uint(P[i + 16]), false, false);
      P[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
  } else {
    const bool upper = (lane & HALF) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const float send = upper ? P[i] : P[i + HALF];
      const float recv = lane_xor<HALF>(send);
      const float keep = upper ? P[i + HALF] : P[i];
      P[i] = keep + recv;
    }
  }
}
template <int HALF, int N>
__device__ __forceinline__ void reduce_step_n(float (&P)[N], int lane) {
  if constexpr (HALF >= N) all_reduce_step<HALF, N>(P);
  else scatter_step_n<HALF, N>(P, lane);
}
template <int N>
__device__ __forceinline__ float reduce_scatter_n(float (&P)[N], int lane) {
  static_assert(N == 4 || N == 8 || N == 16 || N == 32 || N == 64, "N must be a power of two in [4, 64]");
  reduce_step_n<32, N>(P, lane);
  reduce_step_n<16, N>(P, lane);
  reduce_step_n<8, N>(P, lane);
  reduce_step_n<4, N>(P, lane);
  reduce_step_n<2, N>(P, lane);
  reduce_step_n<1, N>(P, lane);
  return P[0];
}

// H heads, G slots per score group (G * H <= 64: one reduce-scatter delivers the group's scores)
template <int H, int G>
__global__ __launch_bounds__(256) void tgat_attn_reduce_kernel(const AttnArgs a) {
  static_assert(G * H <= 64, "a score group must fit the 64 lanes");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int k = a.k, T = a.T, d = a.d, D = a.D;
  // per-wave LDS: cos cache [k][T], dt [k], valid [k], A [H][k]
  float* s_cos = lds_all + (size_t)wave * (k * T + k * (H + 2));
  float* s_dt = s_cos + k * T;
  float* s_valid = s_dt + k;
  float* s_A = s_valid + k;

  const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  if (r >= a.R) return;
  const float* __restrict__ q = a.qf + r * (long long)H * a.Cs;
  const AttnLevel lv = attn_level(a, r);
  const float* __restrict__ nb = lv.nbrf + r * (long long)k * d;
  const float* __restrict__ ex = lv.ex + r * (long long)k * D;

  // slot metadata + Time2Vec of every slot (computed once, reused by both passes)
  const long long st = a.tfeat ? 0 : lv.seed_t[r];
  for (int s = lane; s < k; s += kWave) {
    if (!a.tfeat) s_dt[s] = (float)(st - lv.nbr_t[r * k + s]);  // int64 subtract, then round-to-nearest f32 (tgat.py:143-145)
    const bool ok = a.mask ? a.mask[r * k + s] != 0 : lv.nbr_id[r * k + s] != -1;
    s_valid[s] = ok ? 1.f : 0.f;
  }
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < k * T; e += kWave) {
    const int s = e / T, t = e - s * T;
    // Linear(1,T) is one fma (time_encoding.py:23-24)
    s_cos[e] = a.tfeat ? a.tfeat[r * (long long)k * T + e] : cos_t2v(__fmaf_rn(s_dt[s], a.tw[t], a.tb[t]));
  }
  __builtin_amdgcn_wave_barrier();

  for (int s0 = 0; s0 < k; s0 += G) {
    const int gs = (k - s0) < G ? (k - s0) : G;
    // ---- pass 1: P[s*H + h] = partial of qf[h] . z[s] over this lane's columns ----
    float P[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) P[j] = 0.f;
    for (int c = lane; c < d; c += kWave) {
      float qv[H];
#pragma unroll
      for (int h = 0; h < H; ++h) qv[h] = q[h * a.Cs + c];
#pragma clang loop unroll(full)
      for (int s = 0; s < G; ++s) {
        // no branch: slots past the group's end re-read the last slot (their lanes are discarded),
        // so all G loads are independent and in flight together
        const int sl = s0 + s < k ? s0 + s : k - 1;
        const float z = nb[(long long)sl * d + c];
#pragma unroll
        for (int h = 0; h < H; ++h) P[s * H + h] = __fmaf_rn(qv[h], z, P[s * H + h]);
      }
    }
    for (int c = lane; c < D; c += kWave) {
      float qv[H];
#pragma unroll
      for (int h = 0; h < H; ++h) qv[h] = q[h * a.Cs + d + c];
#pragma clang loop unroll(full)
      for (int s = 0; s < G; ++s) {
        // no branch: slots past the group's end re-read the last slot (their lanes are discarded),
        // so all G loads are independent and in flight together
        const int sl = s0 + s < k ? s0 + s : k - 1;
        const float z = ex[(long long)sl * D + c];
#pragma unroll
        for (int h = 0; h < H; ++h) P[s * H + h] = __fmaf_rn(qv[h], z, P[s * H + h]);
      }
    }
    for (int c = lane; c < T; c += kWave) {
      float qv[H];
#pragma unroll
      for (int h = 0; h < H; ++h) qv[h] = q[h * a.Cs + d + D + c];
#pragma clang loop unroll(full)
      for (int s = 0; s < G; ++s) {
        // no branch: slots past the group's end re-read the last slot (their lanes are discarded),
        // so all G loads are independent and in flight together
        const int sl = s0 + s < k ? s0 + s : k - 1;
        const float z = s_cos[sl * T + c];
#pragma unroll
        for (int h = 0; h < H; ++h) P[s * H + h] = __fmaf_rn(qv[h], z, P[s * H + h]);
      }
    }
    const float sc = reduce_scatter64(P, lane);  // lane j = s*H + h now holds the full dot product
    const int s = lane / H, h = lane - s * H;
    if (s < gs) s_A[h * k + s0 + s] = s_valid[s0 + s] != 0.f ? sc * a.scale : -1e10f;  // masked_fill (attention.py:117)
  }
  __builtin_amdgcn_wave_barrier();

  // ---- softmax over the k slots of every head (attention.py:118) ----
  if (lane < H) {
    float* row = s_A + lane * k;
    float mx = row[0];
    for (int s = 1; s < k; ++s) mx = fmaxf(mx, row[s]);
    float sum = 0.f;
    for (int s = 0; s < k; ++s) {
      const float e = expf(row[s] - mx);
      row[s] = e;
      sum += e;
    }
    for (int s = 0; s < k; ++s) row[s] = row[s] / sum;
  }
  __builtin_amdgcn_wave_barrier();

  if (a.probs)
    for (int e = lane; e < H * k; e += kWave) a.probs[r * (long long)H * k + e] = s_A[e];
  if (a.drop.thresh) {
    for (int e = lane; e < H * k; e += kWave) s_A[e] *= dropout_scale(a.drop, (unsigned long long)(a.drop_row0 + r) * (H * k) + e);
    __builtin_amdgcn_wave_barrier();
  }
  // ---- pass 2: zbar[h][c] = sum_s A[h][s] z[s][c]  (features re-read: L1/L2 hits) ----
  // slots are consumed 8 at a time so that 8 independent loads are in flight per lane
  float* __restrict__ zb = a.zbar + r * (long long)H * a.Cs;
  constexpr int U = 8;
  auto weighted_sum = [&](const float* __restrict__ base, long long slot_stride, int dim, int col0) {
    for (int c = lane; c < dim; c += kWave) {
      float acc[H];
#pragma unroll
      for (int h = 0; h < H; ++h) acc[h] = 0.f;
      for (int s0 = 0; s0 < k; s0 += U) {
        float z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int s = s0 + u < k ? s0 + u : k - 1;  // clamped: the tail is multiplied by 0 below
          z[u] = base[(long long)s * slot_stride + c];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (s0 + u < k) {
#pragma unroll
            for (int h = 0; h < H; ++h) acc[h] = __fmaf_rn(s_A[h * k + s0 + u], z[u], acc[h]);
          }
        }
      }
#pragma unroll
      for (int h = 0; h < H; ++h) zb[h * a.Cs + col0 + c] = acc[h];
    }
  };
  weighted_sum(nb, d, d, 0);
  if (D > 0) weighted_sum(ex, D, D, d);
  weighted_sum(s_cos, T, T, d + D);
}

// ---------------------------------------------------------------------------
// Register-resident variant for the shapes TGAT actually runs (k <= G slots in one score group,
// D a multiple of 4 with D/4 <= 64, T <= 128, times given): every lane keeps "its" float4 column of
// all k slots' edge features (and of the neighbor features) plus its two Time2Vec columns in
// registers, so the row's [k, C] block is read from memory exactly ONCE, as 16-byte loads issued
// back to back; no LDS at all (slot metadata and attention weights travel by v_readlane).
// ---------------------------------------------------------------------------
// value of lane `src` (a compile-time constant after unrolling) for every lane: one v_readlane_b32 into an SGPR --
// __shfl would go through the LDS crossbar (ds_bpermute) even for a constant lane
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Are ALL of a row's Time2Vec arguments fma(dt_s, w_c, b_c) (slots s < G held one per lane in my_dt, this lane's two columns) inside the
// float reduction's range?  The exact test is G x 2 evaluations per lane (140 instructions a row); a sufficient one comes first --
// |dt|_max |w| + |b| is below the limit with 1 % to spare, which no rounding of the fma can make up -- and only a row that fails it (time
// deltas of ~10^6 and more times the largest frequency) takes the exact loop: the answer is the exact test's in every case.
template <int G>
__device__ __forceinline__ bool row_args_small(float my_dt, float w0, float b0, float w1, float b1) {
  float m = fabsf(my_dt);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const float lim = 0.99f * kCosSmallLimit;
  if (__all(__fmaf_rn(m, fabsf(w0), fabsf(b0)) < lim && __fmaf_rn(m, fabsf(w1), fabsf(b1)) < lim)) return true;
  bool small = true;
#pragma unroll
  for (int s = 0; s < G; ++s) {
    const float dt = lane_bcast(my_dt, s);
    small = small && fabsf(__fmaf_rn(dt, w0, b0)) < kCosSmallLimit && fabsf(__fmaf_rn(dt, w1, b1)) < kCosSmallLimit;
  }
  return __all(small);
}

// IDX: compact rows / neighbor features by index (AttnArgs.seg_uniq, seg_live, seg_nidx) -- a variant of its own so that the plain
// kernels keep their register budget (7 VGPRs would cost the k <= 10 variants a wave per SIMD)
template <int H, int G, bool NBV, bool IDX>
__global__ __launch_bounds__(256) void tgat_attn_reduce_reg_kernel(const AttnArgs a) {
  static_assert(G * H <= 64, "a score group must fit the 64 lanes");
  const int lane = lane_id();
  const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const int k = a.k, T = a.T, d = a.d, D = a.D;
  const int D4 = D >> 2, d4 = d >> 2;
  const float* __restrict__ q = a.qlane ? nullptr : a.qf + r * (long long)H * a.Cs;
  const AttnLevel lv = attn_level(a, r);
  if (IDX && lv.live && r - lv.begin >= (long long)*lv.live) return;  // compact rows: past the level's distinct rows
  // ro: the row of the sampler's outputs this row reads (compact rows: the original row it stands for); qf / zbar / probs live at r
  const long long ro = (IDX && lv.uniq) ? (long long)lv.uniq[r] + lv.begin : r;
  const float* __restrict__ nb = lv.nbrf + ro * (long long)k * d;
  const float4* __restrict__ ex4 = reinterpret_cast<const float4*>(lv.ex + ro * (long long)k * D);

  // slot metadata: lane s holds slot s
  float my_dt = 0.f;
  bool my_ok = false;
  int my_eid = -1;  // edge features by id: lane s holds slot s's edge id
  int my_nix = 0;   // neighbor features by index: lane s holds slot s's row of lv.ntab
  if (lane < k) {
    my_dt = (float)(lv.seed_t[ro] - lv.nbr_t[ro * k + lane]);  // int64 subtract, then round-to-nearest f32
    my_ok = a.mask ? a.mask[ro * k + lane] != 0 : lv.nbr_id[ro * k + lane] != -1;
    if (lv.eid) my_eid = lv.eid[ro * k + lane];
    if (IDX && lv.nidx) my_nix = lv.nidx[ro * k + lane];
  }
  // slot sl's neighbor feature row (sl wave-uniform: scalar address arithmetic)
  auto nrow = [&](int sl) __attribute__((always_inline)) -> const float* {
    if (!IDX || !lv.nidx) return nb + (long long)sl * d;
    const int ix = __builtin_amdgcn_readlane(my_nix, sl);
    return lv.ntab + (ix < 0 ? lv.npad : (long long)ix) * d;
  };
  const float4* __restrict__ table4 = reinterpret_cast<const float4*>(lv.table);

  // The sampler's all-pad row (every seed that is itself a pad slot of the hop above: ~1/3 of the layer-1 rows at the
  // headline shape): no valid slot, so the reference attends uniformly (attention.py:114-118) over k slots that are all
  // the same pad entry -- the same neighbor row (node -1), zero edge features (recency.py:287-319 pads with 0.0) and one
  // time delta.  The uniform average of k identical vectors is that vector: no scores, no softmax, ONE slot read instead
  // of k.  (Differs from summing k products by (1/k) in the last bit or two: ~1e-7 relative.)  Applies to id-masked
  // rows only; an explicit mask tensor takes the general path.
  const bool e_on = lane < D4;
  {
    const unsigned kmask = k >= 32 ? 0xffffffffu : ((1u << k) - 1u);
    const bool none = ((unsigned)__ballot(my_ok) & kmask) == 0 && !a.mask;
    const float dt0 = lane_bcast(my_dt, 0);
    if (none && __all(lane >= k || my_dt == dt0)) {
      const float A = 1.0f / (float)k;  // softmax of k equal scores
      if (a.probs && lane < H * k) a.probs[r * (long long)H * k + lane] = A;
      float wh[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        wh[h] = 1.f;
        if (a.drop.thresh) {  // sum over the slots of A * dropout scale
          float acc = 0.f;
          for (int s2 = 0; s2 < k; ++s2) acc += A * dropout_scale(a.drop, (unsigned long long)(a.drop_row0 + r) * (H * k) + h * k + s2);
          wh[h] = acc;
        }
      }
      const bool t0 = lane < T, t1 = lane + kWave < T;
      const float c0 = cos_t2v(__fmaf_rn(dt0, t0 ? a.tw[lane] : 0.f, t0 ? a.tb[lane] : 0.f));
      const float c1 = cos_t2v(__fmaf_rn(dt0, t1 ? a.tw[lane + kWave] : 0.f, t1 ? a.tb[lane + kWave] : 0.f));
      const float4 e = (e_on && !lv.eid) ? ex4[lane] : make_float4(0.f, 0.f, 0.f, 0.f);  // (by id: an all-pad row's slots have no edge)
      float* __restrict__ zb0 = a.zbar + r * (long long)H * a.Cs;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float* zh = zb0 + h * a.Cs;
        const float w = wh[h];
        if (NBV) {
          if (lane < d4) {
            const float4 v = reinterpret_cast<const float4*>(nrow(0))[lane];
            zh[4 * lane] = w * v.x; zh[4 * lane + 1] = w * v.y; zh[4 * lane + 2] = w * v.z; zh[4 * lane + 3] = w * v.w;
          }
        } else if (lane < d) {
          zh[lane] = w * nrow(0)[lane];
        }
        if (e_on) { zh[d + 4 * lane] = w * e.x; zh[d + 4 * lane + 1] = w * e.y; zh[d + 4 * lane + 2] = w * e.z; zh[d + 4 * lane + 3] = w * e.w; }
        if (t0) zh[d + D + lane] = w * c0;
        if (t1) zh[d + D + lane + kWave] = w * c1;
      }
      return;
    }
  }

  // ---- the slots that matter ----
  // A masked slot of a row that has a valid one gets attention weight exactly 0 (exp(-1e10 - max) == 0): it adds 0 to the softmax's
  // sum and fma(0, z, acc) == acc to the weighted mean, so the slots LEFT of the row's first valid one need not be read, scored or
  // summed at all.  The sampler right-aligns a window (pads on the left) and at the headline shape a real row holds ~7 of 20: the body
  // below is instantiated for the last GS = 4, 8, 12, 16 and all G slots and the row takes the smallest that covers its span.  Scores
  // travel back to the lane of their ORIGINAL slot before the softmax, every sum keeps its order: bit-identical to the all-slots body.
  // A row with NO valid slot attends uniformly over all of them (attention.py:114-118): all G.
  const unsigned long long okm = __ballot(my_ok);
  const int span = okm ? k - (__ffsll((long long)okm) - 1) : k;  // slots from the first valid one to the end
  // rows whose k slots all carry the same time delta need ONE Time2Vec evaluation per column, not k -- identical arithmetic
  const bool same_dt = __all(lane >= k || my_dt == lane_bcast(my_dt, 0));
  const bool e_on2 = e_on;
  auto body = [&](auto gsc) __attribute__((always_inline)) {
    constexpr int GS = decltype(gsc)::value;
    constexpr int NV = GS * H, NVp = NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;
    const int s0 = k > GS ? k - GS : 0;  // first slot this body looks at (wave-uniform)
    // ---- the row's features, straight into registers (all loads independent) ----
    float4 ze[GS];
#pragma unroll
    for (int s = 0; s < GS; ++s) {
      const int sl = s0 + s < k ? s0 + s : k - 1;
      if (lv.eid) {  // wave-uniform: the slot's row of the resident store (scalar base + lane offset), zeros for a pad slot
        const int e = __builtin_amdgcn_readlane(my_eid, sl);
        ze[s] = (e_on2 && e >= 0) ? table4[(long long)e * D4 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        ze[s] = e_on2 ? ex4[sl * D4 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 zn[NBV ? GS : 1];
    float zs[NBV ? 1 : GS];
    if (NBV) {
#pragma unroll
      for (int s = 0; s < GS; ++s) {
        const int sl = s0 + s < k ? s0 + s : k - 1;
        zn[NBV ? s : 0] = lane < d4 ? reinterpret_cast<const float4*>(nrow(sl))[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int s = 0; s < GS; ++s) {
        const int sl = s0 + s < k ? s0 + s : k - 1;
        zs[NBV ? 0 : s] = lane < d ? nrow(sl)[lane] : 0.f;
      }
    }
    // folded query columns of this lane
    float4 qe[H], qn4[H];
    float qn[H], qt0[H], qt1[H];
    const bool t0_on = lane < T, t1_on = lane + kWave < T;
    if (a.qlane) {  // wave-uniform: four contiguous 16-byte loads per head from a cache-resident table instead of 7 scattered ones
      const float x = a.qx[r];
      const float4* __restrict__ tab = reinterpret_cast<const float4*>(a.qlane) + lane * 4;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float4 v4 = tab[h * 256], u4 = tab[h * 256 + 1], tt = tab[h * 256 + 2], nn = tab[h * 256 + 3];
        qe[h] = make_float4(__fmaf_rn(x, u4.x, v4.x), __fmaf_rn(x, u4.y, v4.y), __fmaf_rn(x, u4.z, v4.z), __fmaf_rn(x, u4.w, v4.w));
        qt0[h] = __fmaf_rn(x, tt.y, tt.x);
        qt1[h] = __fmaf_rn(x, tt.w, tt.z);
        if (NBV) qn4[h] = make_float4(0.f, 0.f, 0.f, 0.f);  // (d == 1 is never the float4 neighbor variant)
        else qn[h] = __fmaf_rn(x, nn.y, nn.x);
      }
    } else
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float* qh = q + h * a.Cs;
      qe[h] = e_on2 ? make_float4(qh[d + 4 * lane], qh[d + 4 * lane + 1], qh[d + 4 * lane + 2], qh[d + 4 * lane + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (NBV) qn4[h] = lane < d4 ? make_float4(qh[4 * lane], qh[4 * lane + 1], qh[4 * lane + 2], qh[4 * lane + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      else qn[h] = lane < d ? qh[lane] : 0.f;
      qt0[h] = t0_on ? qh[d + D + lane] : 0.f;
      qt1[h] = t1_on ? qh[d + D + lane + kWave] : 0.f;
    }
    const float w0 = t0_on ? a.tw[lane] : 0.f, b0 = t0_on ? a.tb[lane] : 0.f;
    const float w1 = t1_on ? a.tw[lane + kWave] : 0.f, b1 = t1_on ? a.tb[lane + kWave] : 0.f;

    // ---- Time2Vec columns + partial scores ----
    float tz0[GS], tz1[GS], P[NVp];
#pragma unroll
    for (int j = 0; j < NVp; ++j) P[j] = 0.f;
    // The cosine's reduction path (float for |x| < 8e6, double beyond) is chosen ONCE per row (over ALL k slots, whatever GS: the
    // choice must not depend on the body), and the evaluations run as straight-line code: with the choice (a wave vote and a branch)
    // inside every evaluation the 40 evaluations of a row cost 23 of the kernel's 73 us.  Lanes past T carry w = b = 0: their
    // cos(0) meets a zero query weight below and is never stored.
    // A masked slot of a row that has a valid one: its Time2Vec columns never reach the output, so they are not evaluated (zeros
    // stand in).  A row with NO valid slot attends uniformly over all of them: everything is needed.
    const unsigned long long need = okm ? okm : ~0ull;
    const bool row_small = row_args_small<G>(my_dt, w0, b0, w1, b1);
    if (same_dt) {
      const float dt = lane_bcast(my_dt, 0);
      const float c0 = row_small ? cos_t2v_small(__fmaf_rn(dt, w0, b0)) : cos_t2v_big(__fmaf_rn(dt, w0, b0));
      const float c1 = row_small ? cos_t2v_small(__fmaf_rn(dt, w1, b1)) : cos_t2v_big(__fmaf_rn(dt, w1, b1));
#pragma unroll
      for (int s = 0; s < GS; ++s) {
        tz0[s] = c0;
        tz1[s] = c1;
      }
    } else if (row_small) {
#pragma unroll
      for (int s = 0; s < GS; ++s) {
        tz0[s] = tz1[s] = 0.f;
        const int sl = s0 + s < k ? s0 + s : k - 1;
        if ((need >> sl) & 1) {  // wave-uniform
          const float dt = lane_bcast(my_dt, sl);
          tz0[s] = cos_t2v_small(__fmaf_rn(dt, w0, b0));
          tz1[s] = cos_t2v_small(__fmaf_rn(dt, w1, b1));
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < GS; ++s) {
        tz0[s] = tz1[s] = 0.f;
        const int sl = s0 + s < k ? s0 + s : k - 1;
        if ((need >> sl) & 1) {
          const float dt = lane_bcast(my_dt, sl);
          tz0[s] = cos_t2v_big(__fmaf_rn(dt, w0, b0));
          tz1[s] = cos_t2v_big(__fmaf_rn(dt, w1, b1));
        }
      }
    }
#pragma unroll
    for (int s = 0; s < GS; ++s) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float p = qe[h].x * ze[s].x;
        p = __fmaf_rn(qe[h].y, ze[s].y, p);
        p = __fmaf_rn(qe[h].z, ze[s].z, p);
        p = __fmaf_rn(qe[h].w, ze[s].w, p);
        if (NBV) {
          p = __fmaf_rn(qn4[h].x, zn[NBV ? s : 0].x, p);
          p = __fmaf_rn(qn4[h].y, zn[NBV ? s : 0].y, p);
          p = __fmaf_rn(qn4[h].z, zn[NBV ? s : 0].z, p);
          p = __fmaf_rn(qn4[h].w, zn[NBV ? s : 0].w, p);
        } else {
          p = __fmaf_rn(qn[h], zs[NBV ? 0 : s], p);
        }
        p = __fmaf_rn(qt0[h], tz0[s], p);
        p = __fmaf_rn(qt1[h], tz1[s], p);
        P[s * H + h] = p;
      }
    }
    // every lane L ends up with entry L mod NVp (summed over the 64 lanes in the order xor 32, 16, 8, 4, 2, 1 whatever NVp is)
    float sc = reduce_scatter_n<NVp>(P, lane);
    // ... and lane j = slot * H + h of the ORIGINAL numbering takes entry j - s0 * H
    if (GS < G) sc = __shfl(sc, (lane - s0 * H) & (NVp - 1));

    // ---- masked softmax over the slots of each head: lanes j, j +- H, ... share a head ----
    const int js = lane / H;
    const bool live = js < k;
    const bool ok = __shfl(my_ok ? 1 : 0, js < k ? js : 0) != 0;
    sc = live ? (ok ? sc * a.scale : -1e10f) : -__builtin_inff();
    const float mx = butterfly_max<H>(sc);
    float ev = live ? expf(sc - mx) : 0.f;
    const float sum = butterfly_sum<H>(ev);
    float A = ev / sum;
    if (a.probs && live) a.probs[r * (long long)H * k + (lane - js * H) * k + js] = A;
    if (a.drop.thresh && live) A *= dropout_scale(a.drop, (unsigned long long)(a.drop_row0 + r) * (H * k) + (lane - js * H) * k + js);

    // ---- zbar[h] = sum_s A[h][s] z[s], from registers ----
    float* __restrict__ zb = a.zbar + r * (long long)H * a.Cs;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float4 ae = make_float4(0.f, 0.f, 0.f, 0.f), an4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float an = 0.f, at0 = 0.f, at1 = 0.f;
#pragma unroll
      for (int s = 0; s < GS; ++s) {
        // (k < G: the clamped duplicate slots past k carry weight 0 -- lane (s0 + s) * H + h is not live)
        const float w = lane_bcast(A, (s0 + s) * H + h);
        ae.x = __fmaf_rn(w, ze[s].x, ae.x); ae.y = __fmaf_rn(w, ze[s].y, ae.y);
        ae.z = __fmaf_rn(w, ze[s].z, ae.z); ae.w = __fmaf_rn(w, ze[s].w, ae.w);
        if (NBV) {
          an4.x = __fmaf_rn(w, zn[NBV ? s : 0].x, an4.x); an4.y = __fmaf_rn(w, zn[NBV ? s : 0].y, an4.y);
          an4.z = __fmaf_rn(w, zn[NBV ? s : 0].z, an4.z); an4.w = __fmaf_rn(w, zn[NBV ? s : 0].w, an4.w);
        } else {
          an = __fmaf_rn(w, zs[NBV ? 0 : s], an);
        }
        at0 = __fmaf_rn(w, tz0[s], at0);
        at1 = __fmaf_rn(w, tz1[s], at1);
      }
      float* zh = zb + h * a.Cs;
      if (NBV) {
        if (lane < d4) { zh[4 * lane] = an4.x; zh[4 * lane + 1] = an4.y; zh[4 * lane + 2] = an4.z; zh[4 * lane + 3] = an4.w; }
      } else if (lane < d) {
        zh[lane] = an;
      }
      if (e_on2) { zh[d + 4 * lane] = ae.x; zh[d + 4 * lane + 1] = ae.y; zh[d + 4 * lane + 2] = ae.z; zh[d + 4 * lane + 3] = ae.w; }
      if (t0_on) zh[d + D + lane] = at0;
      if (t1_on) zh[d + D + lane + kWave] = at1;
    }
  };
  // (a.full_span: the A/B knob TGMX_ATTN_SPAN=0 -- every row through the all-slots body)
  const int want = a.full_span ? G : span;
  if constexpr (G > 16) {
    if (want <= 4) return body(std::integral_constant<int, 4>{});
    if (want <= 8) return body(std::integral_constant<int, 8>{});
    if (want <= 12) return body(std::integral_constant<int, 12>{});
    if (want <= 16) return body(std::integral_constant<int, 16>{});
  } else if constexpr (G > 8) {
    if (want <= 4) return body(std::integral_constant<int, 4>{});
    if (want <= 8) return body(std::integral_constant<int, 8>{});
  }
  body(std::integral_constant<int, G>{});
}

// ---------------------------------------------------------------------------
// The same row on FOUR waves (one workgroup per row): for launches of few rows (the 600-row layer of the headline forward: 600 waves
// on 1024 SIMDs, so the launch lasts as long as ONE 20-slot row -- 14.6 us).
// Wave w takes slots [w Q, (w + 1) Q), Q = ceil(G / 4): loads, Time2Vec and partial scores of its slots only (a quarter of the
// register-resident body's registers and of its serial work).  BIT-IDENTICAL to tgat_attn_reduce_reg_kernel by construction:
//   * a score is summed over the 64 lanes in the order xor 32, 16, 8, 4, 2, 1 whatever the number of entries (reduce_scatter_n);
//   * the scores meet in LDS and EVERY wave runs the same softmax butterfly over the k slots;
//   * the weighted mean is one fma chain over the slots in ascending order: the waves take turns, the running sums travel through
//     LDS from wave w to wave w + 1 (three hand-offs of 10 H floats per lane), the last wave stores;
//   * waves whose slots all lie left of the row's first valid slot end at once (their slots carry weight exactly 0, see the span
//     bodies above) -- a wave that has ended is not counted by s_barrier.
// No compact rows, attention-weight output or dropout here: the launch code sends those to the one-wave kernel.
// ---------------------------------------------------------------------------
template <int H, int G, bool NBV>
__global__ __launch_bounds__(256) void tgat_attn_reduce_mw_kernel(const AttnArgs a) {
  constexpr int W = 4, Q = (G + W - 1) / W;
  constexpr int NV = Q * H, NVp = NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;
  static_assert(G * H <= 64, "a score group must fit the 64 lanes");
  __shared__ float sc_lds[64];
  __shared__ float acc_lds[H * 10][kWave];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const long long r = blockIdx.x;
  const int k = a.k, T = a.T, d = a.d, D = a.D;
  const int D4 = D >> 2, d4 = d >> 2;
  const float* __restrict__ q = a.qlane ? nullptr : a.qf + r * (long long)H * a.Cs;
  const AttnLevel lv = attn_level(a, r);
  const float* __restrict__ nb = lv.nbrf + r * (long long)k * d;
  const float4* __restrict__ ex4 = reinterpret_cast<const float4*>(lv.ex + r * (long long)k * D);
  float my_dt = 0.f;
  bool my_ok = false;
  int my_eid = -1;
  if (lane < k) {
    my_dt = (float)(lv.seed_t[r] - lv.nbr_t[r * k + lane]);
    my_ok = lv.nbr_id[r * k + lane] != -1;
    if (lv.eid) my_eid = lv.eid[r * k + lane];
  }
  const float4* __restrict__ table4 = reinterpret_cast<const float4*>(lv.table);
  const bool e_on = lane < D4;
  const bool t0_on = lane < T, t1_on = lane + kWave < T;
  const unsigned long long okm = __ballot(my_ok);
  const bool same_dt = __all(lane >= k || my_dt == lane_bcast(my_dt, 0));
  if (!okm && same_dt) {  // the sampler's all-pad row: the one-wave kernel's shortcut, by wave 0 alone (same arithmetic)
    if (wave != 0) return;
    const float dt0 = lane_bcast(my_dt, 0);
    const float c0 = cos_t2v(__fmaf_rn(dt0, t0_on ? a.tw[lane] : 0.f, t0_on ? a.tb[lane] : 0.f));
    const float c1 = cos_t2v(__fmaf_rn(dt0, t1_on ? a.tw[lane + kWave] : 0.f, t1_on ? a.tb[lane + kWave] : 0.f));
    const float4 e = (e_on && !lv.eid) ? ex4[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    float* __restrict__ zb0 = a.zbar + r * (long long)H * a.Cs;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float* zh = zb0 + h * a.Cs;
      const float w = 1.f;
      if (NBV) {
        if (lane < d4) {
          const float4 v = reinterpret_cast<const float4*>(nb)[lane];
          zh[4 * lane] = w * v.x; zh[4 * lane + 1] = w * v.y; zh[4 * lane + 2] = w * v.z; zh[4 * lane + 3] = w * v.w;
        }
      } else if (lane < d) {
        zh[lane] = w * nb[lane];
      }
      if (e_on) { zh[d + 4 * lane] = w * e.x; zh[d + 4 * lane + 1] = w * e.y; zh[d + 4 * lane + 2] = w * e.z; zh[d + 4 * lane + 3] = w * e.w; }
      if (t0_on) zh[d + D + lane] = w * c0;
      if (t1_on) zh[d + D + lane + kWave] = w * c1;
    }
    return;
  }
  const int span = okm ? k - (__ffsll((long long)okm) - 1) : k;
  const int w_first = (k - span) / Q;  // the first wave that owns a slot that matters (workgroup-uniform)
  if (wave < w_first) return;
  const int s_lo = wave * Q;

  // ---- this wave's slots, straight into registers ----
  float4 ze[Q];
#pragma unroll
  for (int s = 0; s < Q; ++s) {
    const int sl = s_lo + s < k ? s_lo + s : k - 1;
    if (lv.eid) {
      const int e = __builtin_amdgcn_readlane(my_eid, sl);
      ze[s] = (e_on && e >= 0) ? table4[(long long)e * D4 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      ze[s] = e_on ? ex4[sl * D4 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 zn[NBV ? Q : 1];
  float zs[NBV ? 1 : Q];
#pragma unroll
  for (int s = 0; s < Q; ++s) {
    const int sl = s_lo + s < k ? s_lo + s : k - 1;
    if (NBV) zn[NBV ? s : 0] = lane < d4 ? reinterpret_cast<const float4*>(nb + (long long)sl * d)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    else zs[NBV ? 0 : s] = lane < d ? nb[(long long)sl * d + lane] : 0.f;
  }
  float4 qe[H], qn4[H];
  float qn[H], qt0[H], qt1[H];
  if (a.qlane) {
    const float x = a.qx[r];
    const float4* __restrict__ tab = reinterpret_cast<const float4*>(a.qlane) + lane * 4;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float4 v4 = tab[h * 256], u4 = tab[h * 256 + 1], tt = tab[h * 256 + 2], nn = tab[h * 256 + 3];
      qe[h] = make_float4(__fmaf_rn(x, u4.x, v4.x), __fmaf_rn(x, u4.y, v4.y), __fmaf_rn(x, u4.z, v4.z), __fmaf_rn(x, u4.w, v4.w));
      qt0[h] = __fmaf_rn(x, tt.y, tt.x);
      qt1[h] = __fmaf_rn(x, tt.w, tt.z);
      if (NBV) qn4[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      else qn[h] = __fmaf_rn(x, nn.y, nn.x);
    }
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float* qh = q + h * a.Cs;
      qe[h] = e_on ? make_float4(qh[d + 4 * lane], qh[d + 4 * lane + 1], qh[d + 4 * lane + 2], qh[d + 4 * lane + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (NBV) qn4[h] = lane < d4 ? make_float4(qh[4 * lane], qh[4 * lane + 1], qh[4 * lane + 2], qh[4 * lane + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      else qn[h] = lane < d ? qh[lane] : 0.f;
      qt0[h] = t0_on ? qh[d + D + lane] : 0.f;
      qt1[h] = t1_on ? qh[d + D + lane + kWave] : 0.f;
    }
  }
  const float w0 = t0_on ? a.tw[lane] : 0.f, b0 = t0_on ? a.tb[lane] : 0.f;
  const float w1 = t1_on ? a.tw[lane + kWave] : 0.f, b1 = t1_on ? a.tb[lane + kWave] : 0.f;

  // ---- Time2Vec columns + partial scores of this wave's slots (the cosine's reduction path: chosen over ALL k slots, as there) ----
  float tz0[Q], tz1[Q], P[NVp];
#pragma unroll
  for (int j = 0; j < NVp; ++j) P[j] = 0.f;
  const unsigned long long need = okm ? okm : ~0ull;
  const bool row_small = row_args_small<G>(my_dt, w0, b0, w1, b1);
  if (same_dt) {
    const float dt = lane_bcast(my_dt, 0);
    const float c0 = row_small ? cos_t2v_small(__fmaf_rn(dt, w0, b0)) : cos_t2v_big(__fmaf_rn(dt, w0, b0));
    const float c1 = row_small ? cos_t2v_small(__fmaf_rn(dt, w1, b1)) : cos_t2v_big(__fmaf_rn(dt, w1, b1));
#pragma unroll
    for (int s = 0; s < Q; ++s) {
      tz0[s] = c0;
      tz1[s] = c1;
    }
  } else {
#pragma unroll
    for (int s = 0; s < Q; ++s) {
      tz0[s] = tz1[s] = 0.f;
      const int sl = s_lo + s < k ? s_lo + s : k - 1;
      if ((need >> sl) & 1) {  // wave-uniform
        const float dt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_dt), sl));
        if (row_small) {
          tz0[s] = cos_t2v_small(__fmaf_rn(dt, w0, b0));
          tz1[s] = cos_t2v_small(__fmaf_rn(dt, w1, b1));
        } else {
          tz0[s] = cos_t2v_big(__fmaf_rn(dt, w0, b0));
          tz1[s] = cos_t2v_big(__fmaf_rn(dt, w1, b1));
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < Q; ++s) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float p = qe[h].x * ze[s].x;
      p = __fmaf_rn(qe[h].y, ze[s].y, p);
      p = __fmaf_rn(qe[h].z, ze[s].z, p);
      p = __fmaf_rn(qe[h].w, ze[s].w, p);
      if (NBV) {
        p = __fmaf_rn(qn4[h].x, zn[NBV ? s : 0].x, p);
        p = __fmaf_rn(qn4[h].y, zn[NBV ? s : 0].y, p);
        p = __fmaf_rn(qn4[h].z, zn[NBV ? s : 0].z, p);
        p = __fmaf_rn(qn4[h].w, zn[NBV ? s : 0].w, p);
      } else {
        p = __fmaf_rn(qn[h], zs[NBV ? 0 : s], p);
      }
      p = __fmaf_rn(qt0[h], tz0[s], p);
      p = __fmaf_rn(qt1[h], tz1[s], p);
      P[s * H + h] = p;
    }
  }
  {
    const float part = reduce_scatter_n<NVp>(P, lane);  // lane L: entry L mod NVp = (slot s_lo + e / H, head e % H)
    if (lane < NV && s_lo * H + lane < 64) sc_lds[s_lo * H + lane] = part;
  }
  __syncthreads();
  // ---- masked softmax over the slots of each head, every wave the same (entries of ended waves: masked slots, never used) ----
  const int js = lane / H;
  const bool live = js < k;
  const bool ok = __shfl(my_ok ? 1 : 0, js < k ? js : 0) != 0;
  float sc = sc_lds[lane];
  sc = live ? (ok ? sc * a.scale : -1e10f) : -__builtin_inff();
  const float mx = butterfly_max<H>(sc);
  const float ev = live ? expf(sc - mx) : 0.f;
  const float sum = butterfly_sum<H>(ev);
  const float A = ev / sum;

  // ---- zbar[h] = sum_s A[h][s] z[s]: one chain over the slots, the waves in turn ----
  float* __restrict__ zb = a.zbar + r * (long long)H * a.Cs;
  for (int w = w_first; w < W; ++w) {
    if (wave == w) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float4 ae = make_float4(0.f, 0.f, 0.f, 0.f), an4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float an = 0.f, at0 = 0.f, at1 = 0.f;
        if (w > w_first) {
          const float(*src)[kWave] = acc_lds + h * 10;
          ae = make_float4(src[0][lane], src[1][lane], src[2][lane], src[3][lane]);
          if (NBV) an4 = make_float4(src[4][lane], src[5][lane], src[6][lane], src[7][lane]);
          else an = src[4][lane];
          at0 = src[8][lane];
          at1 = src[9][lane];
        }
#pragma unroll
        for (int s = 0; s < Q; ++s) {
          const int idx = (s_lo + s) * H + h;  // lanes past k * H hold weight 0
          const float wgt = idx < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A), idx < 64 ? idx : 0)) : 0.f;
          ae.x = __fmaf_rn(wgt, ze[s].x, ae.x); ae.y = __fmaf_rn(wgt, ze[s].y, ae.y);
          ae.z = __fmaf_rn(wgt, ze[s].z, ae.z); ae.w = __fmaf_rn(wgt, ze[s].w, ae.w);
          if (NBV) {
            an4.x = __fmaf_rn(wgt, zn[NBV ? s : 0].x, an4.x); an4.y = __fmaf_rn(wgt, zn[NBV ? s : 0].y, an4.y);
            an4.z = __fmaf_rn(wgt, zn[NBV ? s : 0].z, an4.z); an4.w = __fmaf_rn(wgt, zn[NBV ? s : 0].w, an4.w);
          } else {
            an = __fmaf_rn(wgt, zs[NBV ? 0 : s], an);
          }
          at0 = __fmaf_rn(wgt, tz0[s], at0);
          at1 = __fmaf_rn(wgt, tz1[s], at1);
        }
        if (w + 1 < W) {
          float(*dst)[kWave] = acc_lds + h * 10;
          dst[0][lane] = ae.x; dst[1][lane] = ae.y; dst[2][lane] = ae.z; dst[3][lane] = ae.w;
          if (NBV) { dst[4][lane] = an4.x; dst[5][lane] = an4.y; dst[6][lane] = an4.z; dst[7][lane] = an4.w; }
          else dst[4][lane] = an;
          dst[8][lane] = at0;
          dst[9][lane] = at1;
        } else {
          float* zh = zb + h * a.Cs;
          if (NBV) {
            if (lane < d4) { zh[4 * lane] = an4.x; zh[4 * lane + 1] = an4.y; zh[4 * lane + 2] = an4.z; zh[4 * lane + 3] = an4.w; }
          } else if (lane < d) {
            zh[lane] = an;
          }
          if (e_on) { zh[d + 4 * lane] = ae.x; zh[d + 4 * lane + 1] = ae.y; zh[d + 4 * lane + 2] = ae.z; zh[d + 4 * lane + 3] = ae.w; }
          if (t0_on) zh[d + D + lane] = at0;
          if (t1_on) zh[d + D + lane + kWave] = at1;
        }
      }
    }
    if (w + 1 < W) __syncthreads();
  }
}

template <int H, int G>
static void launch_attn(dim3 grid, dim3 block, size_t lds, hipStream_t st, const AttnArgs& a) {
  if constexpr (G * H <= 64) hipLaunchKernelGGL((tgat_attn_reduce_kernel<H, G>), grid, block, lds, st, a);
}

// register-resident fast path; returns false when the shape does not qualify
template <int H>
static bool launch_attn_reg(hipStream_t st, const AttnArgs& a) {
  uintptr_t ex_bits = (uintptr_t)a.ex, nb_bits = (uintptr_t)a.nbrf;  // D, d multiples of 4: biased pointers keep their alignment
  for (int i = 0; i < a.n_seg; ++i) {
    ex_bits |= (uintptr_t)a.seg_ex[i];
    nb_bits |= a.seg_nidx[i] ? (uintptr_t)a.seg_ntab[i] : (uintptr_t)a.seg_nbrf[i];
  }
  const bool edge_ok = a.D > 0 && a.D % 4 == 0 && a.D / 4 <= 64 && (ex_bits & 15) == 0;
  const bool nbv = a.d % 4 == 0 && a.d / 4 <= 64 && (nb_bits & 15) == 0;
  if (!edge_ok || a.tfeat || a.T > 128 || !(nbv || a.d <= 64)) return false;
  // ONE wave per workgroup (TGMX_ATTN_WPB, 1..4, for A/B runs): the kernel shares nothing between the waves of a workgroup, and a
  // 4-wave workgroup only retires -- and its four wave slots only refill -- when its SLOWEST row is done.  Per-row clock stamps
  // (round 4) showed under half of the 2048 resident slots busy on average with 4-wave workgroups; with 1-wave workgroups a slot
  // refills as soon as its own row ends: the 12 600-row launch 56.5 -> 49.6 us.
  static const int wpb = [] {
    const char* e = getenv("TGMX_ATTN_WPB");
    const int v = e ? atoi(e) : 1;
    return v >= 1 && v <= 4 ? v : 1;
  }();
  const dim3 grid((unsigned)((a.R + wpb - 1) / wpb)), block(64 * wpb);
  bool idx = false;
  for (int i = 0; i < TGMX_TGAT_MAX_LAYERS; ++i) idx |= a.seg_uniq[i] || a.seg_live[i] || a.seg_nidx[i];
  // few rows: four waves per row (tgat_attn_reduce_mw_kernel; TGMX_ATTN_MW=0: the A/B knob)
  static const bool mw_knob = [] { const char* e = getenv("TGMX_ATTN_MW"); return !(e && e[0] == '0'); }();
  const bool mw = mw_knob && a.R <= 2048 && !idx && !a.probs && !a.drop.thresh && !a.mask;
#define TGMX_REG(G_)                                                                                        \
  if constexpr (G_ * H <= 64) {                                                                             \
    if (a.k <= G_) {                                                                                        \
      if (mw) {                                                                                             \
        const dim3 mgrid((unsigned)a.R), mblock(256);                                                       \
        if (nbv) hipLaunchKernelGGL((tgat_attn_reduce_mw_kernel<H, G_, true>), mgrid, mblock, 0, st, a);    \
        else hipLaunchKernelGGL((tgat_attn_reduce_mw_kernel<H, G_, false>), mgrid, mblock, 0, st, a);       \
      } else if (idx) {                                                                                            \
        if (nbv) hipLaunchKernelGGL((tgat_attn_reduce_reg_kernel<H, G_, true, true>), grid, block, 0, st, a);  \
        else hipLaunchKernelGGL((tgat_attn_reduce_reg_kernel<H, G_, false, true>), grid, block, 0, st, a);     \
      } else if (nbv) {                                                                                     \
        hipLaunchKernelGGL((tgat_attn_reduce_reg_kernel<H, G_, true, false>), grid, block, 0, st, a);       \
      } else {                                                                                              \
        hipLaunchKernelGGL((tgat_attn_reduce_reg_kernel<H, G_, false, false>), grid, block, 0, st, a);      \
      }                                                                                                     \
      return true;                                                                                          \
    }                                                                                                       \
  }
  TGMX_REG(10)
  TGMX_REG(20)
#undef TGMX_REG
  return false;
}

}  // namespace tgmx

using namespace tgmx;

// Which problems take the LDS-staged kernel: M >= 6144 rows.  A function of M ALONE, like the few-row kernel's M <= 2048: the levels of a
// hop tree that share a layer's weights run as one row batch in one composition and level by level in another (12 600 = 600 + 12 000
// rows: tests/test_tgat_backward_gpu.py compares the two bit for bit), and a threshold on the number of 64 x 64 blocks -- which fits the
// measurements a little better (8 118 x 172 x 172, 381 blocks: 13.4 us direct / 14.0 staged; 8 118 x 200 x 116, 508 blocks: 14.0 / 11.2)
// -- put 12 000 x 51 x 273 x 2 heads and 12 600 x 51 x 273 x 2 heads on different sides.  Below ~6 k rows a launch is at most 1.5 rounds of
// blocks per CU and the K-split kernel's four short chains per 32 x 64 tile finish sooner (4 100 x 172 x 172: 8.3 / 10.4 us).
constexpr long long kGemmLdsMinRows = 6144;
static bool gemm_takes_lds(long long M, int /*N*/, int /*batch*/) { return M >= kGemmLdsMinRows; }

extern "C" int tgmx_sgemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                             int32_t N, int32_t K, const float* bias, int32_t relu, int32_t batch, int64_t strideA,
                             int64_t strideB, int64_t strideC, tgmx_stream_t stream) {
  TGMX_REQUIRE(M >= 0 && N > 0 && K > 0 && batch > 0, "sgemm_nt: bad sizes M=%lld N=%d K=%d batch=%d", (long long)M, N, K, batch);
  if (M == 0) return TGMX_OK;
  TGMX_REQUIRE(A && B && C, "sgemm_nt: null pointer");
  TGMX_REQUIRE(lda >= K && ldb >= K && ldc >= N, "sgemm_nt: leading dimension smaller than the row length");
  GemmArgs g{A, B, C, bias, lda, ldb, ldc, strideA, strideB, strideC, M, N, K, relu};
  auto vec_ok = [](const float* p, long long ld, long long stride) { return ((uintptr_t)p & 15) == 0 && ld % 4 == 0 && stride % 4 == 0; };
  const bool av = vec_ok(A, lda, strideA), bv = vec_ok(B, ldb, strideB);
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(256);
  // K-splitting across the 4 waves of a block (4x the waves, each with a quarter of the dependent MFMA chain) wins
  // whenever there is more than one 16-wide k-step to hand out; measured on every GEMM shape of the TGAT path
  // (600 .. 12 600 rows, K = 102 .. 448: 1.1x .. 2.9x) and neutral at 4096^3.
  static const bool small_knob = [] { const char* e = getenv("TGMX_GEMM_SMALL"); return !(e && e[0] == '0'); }();  // A/B knob
  if (small_knob && M <= 2048 && K > 16 && K <= 512) {  // (tried up to 64 k rows on the TGN pipeline's GEMMs: no gain past ~2 k)  // few rows: the latency-shaped kernel (see sgemm_nt_small_kernel)
    const dim3 sgrid((unsigned)((M + 31) / 32), (unsigned)((N + 31) / 32), (unsigned)batch), sblock(512);
    const int steps = (K + 127) / 128;
#define TGMX_GEMM_S(AV_, BV_)                                                                              \
  do {                                                                                                     \
    if (steps == 1) hipLaunchKernelGGL((sgemm_nt_small_kernel<AV_, BV_, 1>), sgrid, sblock, 0, st, g);      \
    else if (steps == 2) hipLaunchKernelGGL((sgemm_nt_small_kernel<AV_, BV_, 2>), sgrid, sblock, 0, st, g); \
    else if (steps == 3) hipLaunchKernelGGL((sgemm_nt_small_kernel<AV_, BV_, 3>), sgrid, sblock, 0, st, g); \
    else hipLaunchKernelGGL((sgemm_nt_small_kernel<AV_, BV_, 4>), sgrid, sblock, 0, st, g);                 \
  } while (0)
    if (av && bv) TGMX_GEMM_S(true, true);
    else if (av) TGMX_GEMM_S(true, false);
    else if (bv) TGMX_GEMM_S(false, true);
    else TGMX_GEMM_S(false, false);
#undef TGMX_GEMM_S
    TGMX_CHECK_LAUNCH("sgemm_nt(small)");
    return TGMX_OK;
  }
  static const bool lds_knob = [] { const char* e = getenv("TGMX_GEMM_LDS"); return !(e && e[0] == '0'); }();  // A/B knob: 0 = the direct kernel
  if (lds_knob && K > 16 && gemm_takes_lds(M, N, batch)) {  // many rows: whole lines through LDS (see sgemm_nt_lds_kernel)
    const dim3 lgrid((unsigned)((M + kLdsBM - 1) / kLdsBM), (unsigned)((N + kLdsBN - 1) / kLdsBN), (unsigned)batch);
    if (av && bv) hipLaunchKernelGGL((sgemm_nt_lds_kernel<true, true>), lgrid, block, 0, st, g);
    else if (av) hipLaunchKernelGGL((sgemm_nt_lds_kernel<true, false>), lgrid, block, 0, st, g);
    else if (bv) hipLaunchKernelGGL((sgemm_nt_lds_kernel<false, true>), lgrid, block, 0, st, g);
    else hipLaunchKernelGGL((sgemm_nt_lds_kernel<false, false>), lgrid, block, 0, st, g);
    TGMX_CHECK_LAUNCH("sgemm_nt(lds)");
    return TGMX_OK;
  }
  const bool split = K > 16;
  const dim3 grid = split ? dim3((unsigned)((M + 31) / 32), (unsigned)((N + 63) / 64), (unsigned)batch)
                          : dim3((unsigned)((M + 127) / 128), (unsigned)((N + 63) / 64), (unsigned)batch);
#define TGMX_GEMM(AV_, BV_) \
  do { \
    if (split) hipLaunchKernelGGL((sgemm_nt_kernel<AV_, BV_, true, 8>), grid, block, 0, st, g); \
    else hipLaunchKernelGGL((sgemm_nt_kernel<AV_, BV_, false, 8>), grid, block, 0, st, g); \
  } while (0)
  if (av && bv) TGMX_GEMM(true, true);
  else if (av) TGMX_GEMM(true, false);
  else if (bv) TGMX_GEMM(false, true);
  else TGMX_GEMM(false, false);
#undef TGMX_GEMM
  TGMX_CHECK_LAUNCH("sgemm_nt");
  return TGMX_OK;
}

// internal (csrc/tgn.hip): two independent C = A B^T (+ bias) problems as ONE launch of the K-split kernel; falls back to two launches
// when a problem takes another kernel on its own (few rows, K <= 16, unaligned operands) so that results never depend on the pairing.
int tgmx_internal_sgemm_nt_pair(const GemmCall& c0, const GemmCall& c1, tgmx_stream_t stream) {
  auto solo = [&](const GemmCall& c) {
    return tgmx_sgemm_nt(c.A, c.lda, c.B, c.ldb, c.C, c.ldc, c.M, c.N, c.K, c.bias, c.relu, c.batch, c.sA, c.sB, c.sC, stream);
  };
  auto vec_ok = [](const float* p, long long ld, long long stride) { return ((uintptr_t)p & 15) == 0 && ld % 4 == 0 && stride % 4 == 0; };
  static const bool small_knob = [] { const char* e = getenv("TGMX_GEMM_SMALL"); return !(e && e[0] == '0'); }();
  static const bool pair_knob = [] { const char* e = getenv("TGMX_GEMM_PAIR"); return !(e && e[0] == '0'); }();  // A/B knob
  auto pairable = [&](const GemmCall& c) {
    return c.M > 0 && c.K > 16 && !(small_knob && c.M <= 2048 && c.K <= 512) && vec_ok(c.A, c.lda, c.sA) && vec_ok(c.B, c.ldb, c.sB) && c.batch > 0;
  };
  if (!pair_knob || !pairable(c0) || !pairable(c1)) {
    if (int rc = solo(c0)) return rc;
    return solo(c1);
  }
  GemmPairArgs p;
  const GemmCall* cs[2] = {&c0, &c1};
  for (int q = 0; q < 2; ++q) {
    const GemmCall& c = *cs[q];
    TGMX_REQUIRE(c.A && c.B && c.C && c.N > 0 && c.lda >= c.K && c.ldb >= c.K && c.ldc >= c.N, "sgemm_nt_pair: bad problem %d", q);
    p.g[q] = GemmArgs{c.A, c.B, c.C, c.bias, c.lda, c.ldb, c.ldc, c.sA, c.sB, c.sC, c.M, c.N, c.K, c.relu};
    p.ny[q] = (unsigned)((c.N + 63) / 64);
    p.nz[q] = (unsigned)c.batch;
    p.lds[q] = 0;
  }
  // a pairable problem (aligned, K > 16) takes the same kernel as alone: the LDS-staged one if gemm_takes_lds (unless TGMX_GEMM_LDS=0)
  static const bool lds_knob = [] { const char* e = getenv("TGMX_GEMM_LDS"); return !(e && e[0] == '0'); }();
  const bool l0 = gemm_takes_lds(c0.M, c0.N, c0.batch), l1 = gemm_takes_lds(c1.M, c1.N, c1.batch);
  if (lds_knob && l0 && l1) {
    p.tiles0 = (unsigned)((c0.M + kLdsBM - 1) / kLdsBM);
    const unsigned lt1 = (unsigned)((c1.M + kLdsBM - 1) / kLdsBM);
    const dim3 lgrid(p.tiles0 + lt1, p.ny[0] > p.ny[1] ? p.ny[0] : p.ny[1], p.nz[0] > p.nz[1] ? p.nz[0] : p.nz[1]);
    hipLaunchKernelGGL(sgemm_nt_lds_pair_kernel, lgrid, dim3(256), 0, (hipStream_t)stream, p);
    TGMX_CHECK_LAUNCH("sgemm_nt_pair(lds)");
    return TGMX_OK;
  }
  if (lds_knob && (l0 || l1)) {  // mixed: each through its own body, one launch
    p.lds[0] = l0; p.lds[1] = l1;
    const unsigned t0 = (unsigned)(l0 ? (c0.M + kLdsBM - 1) / kLdsBM : (c0.M + 31) / 32), t1 = (unsigned)(l1 ? (c1.M + kLdsBM - 1) / kLdsBM : (c1.M + 31) / 32);
    p.tiles0 = t0;
    const dim3 mgrid(t0 + t1, p.ny[0] > p.ny[1] ? p.ny[0] : p.ny[1], p.nz[0] > p.nz[1] ? p.nz[0] : p.nz[1]);
    hipLaunchKernelGGL(sgemm_nt_mixed_pair_kernel, mgrid, dim3(256), 0, (hipStream_t)stream, p);
    TGMX_CHECK_LAUNCH("sgemm_nt_pair(mixed)");
    return TGMX_OK;
  }
  p.tiles0 = (unsigned)((c0.M + 31) / 32);
  const unsigned tiles1 = (unsigned)((c1.M + 31) / 32);
  const dim3 grid(p.tiles0 + tiles1, p.ny[0] > p.ny[1] ? p.ny[0] : p.ny[1], p.nz[0] > p.nz[1] ? p.nz[0] : p.nz[1]);
  hipLaunchKernelGGL(sgemm_nt_pair_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  TGMX_CHECK_LAUNCH("sgemm_nt_pair");
  return TGMX_OK;
}

extern "C" int tgmx_gather_rows(const float* table, int64_t num_rows, int32_t dim, const int32_t* idx, int64_t n, float* out,
                                int64_t ldo, tgmx_stream_t stream) {
  TGMX_REQUIRE(num_rows > 0 && dim > 0 && n >= 0 && ldo >= dim, "gather_rows: bad sizes");
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(table && idx && out, "gather_rows: null pointer");
  long long blocks = (n * dim + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, (long long)num_rows,
                     dim, idx, (long long)n, out, (long long)ldo);
  TGMX_CHECK_LAUNCH("gather_rows");
  return TGMX_OK;
}

extern "C" int tgmx_tgat_rres(const float* x, int64_t ldx, int32_t d, const float* tb, const float* time_feat, int32_t T,
                              int32_t O, int64_t R, float* out, int64_t ldo, tgmx_stream_t stream) {
  TGMX_REQUIRE(d > 0 && T > 0 && O >= d + T && R >= 0 && ldx >= d, "tgat_rres: bad sizes d=%d T=%d O=%d", d, T, O);
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(x && (tb || time_feat) && out, "tgat_rres: null pointer");
  if (ldo == 0) ldo = O;
  TGMX_REQUIRE(ldo >= O, "tgat_rres: ldo < O");
  long long blocks = (R * ldo + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tgat_rres_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, d, tb, time_feat,
                     T, O, (long long)R, out, (long long)ldo);
  TGMX_CHECK_LAUNCH("tgat_rres");
  return TGMX_OK;
}

// res == nullptr: the residual is [x[:, :d] | 0 | cos(tb)], rebuilt inside the kernel (no tgat_rres launch)
static int ln_residual_concat_impl(const float* y, int64_t ldy, const float* res, int64_t ldr, const float* gamma, const float* beta,
                                   int32_t O, float eps, const float* z0, int32_t d0, int64_t R, float* out, int64_t ldo, const float* x,
                                   int64_t ldx, int32_t d, const float* tb, int32_t T, tgmx_stream_t stream) {
  TGMX_REQUIRE(O > 0 && d0 >= 0 && R >= 0, "ln_residual_concat: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(y && (res || (x && tb && d >= 0 && T >= 0 && d + T <= O)) && gamma && beta && out && (d0 == 0 || z0), "ln_residual_concat: null pointer");
  if (ldy == 0) ldy = O;
  if (ldr == 0) ldr = O;
  if (ldo == 0) ldo = O + d0;
  TGMX_REQUIRE(ldy >= O && ldr >= O && ldo >= O + d0, "ln_residual_concat: leading dimension too small");
  hipLaunchKernelGGL(ln_residual_concat_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, res, gamma,
                     beta, O, eps, z0, d0, (long long)R, out, (long long)ldy, (long long)ldr, (long long)ldo, x, (long long)ldx, d, tb, T);
  TGMX_CHECK_LAUNCH("ln_residual_concat");
  return TGMX_OK;
}

extern "C" int tgmx_ln_residual_concat(const float* y, int64_t ldy, const float* res, int64_t ldr, const float* gamma,
                                       const float* beta, int32_t O, float eps, const float* z0, int32_t d0, int64_t R, float* out,
                                       int64_t ldo, tgmx_stream_t stream) {
  TGMX_REQUIRE(res, "ln_residual_concat: null pointer");
  return ln_residual_concat_impl(y, ldy, res, ldr, gamma, beta, O, eps, z0, d0, R, out, ldo, nullptr, 0, 0, nullptr, 0, stream);
}

// qf[r, i] = fma(x[r, d-1], U[i, d-1], ... fma(x[r, 0], U[i, 0], v[i]))  for d <= 4 (U rows padded to 4): the folded queries of a layer
// with a narrow input as a streaming outer product, at the write bandwidth instead of MFMA tiles that are 15/16 padding
__global__ __launch_bounds__(256) void tgat_qfold_small_kernel(const float* __restrict__ x, long long ldx, int d, const float* __restrict__ U,
                                                               const float* __restrict__ v, int n4, long long R, float* __restrict__ out,
                                                               long long ldo, const RowSegs segs) {
  // one work item = 4 consecutive columns x 16 consecutive rows: the 4 U rows and v stay in registers, consecutive threads write
  // consecutive 16-byte pieces of a row
  constexpr int RB = 16;
  const long long items = ((R + RB - 1) / RB) * n4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < items; e += (long long)gridDim.x * blockDim.x) {
    const long long rb = e / n4;
    const int i = (int)(e - rb * n4) * 4;
    const float4 vv = *reinterpret_cast<const float4*>(v + i);
    float4 w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const float4*>(U + (long long)(i + u) * 4);
    const long long r1 = (rb + 1) * RB < R ? (rb + 1) * RB : R;
    if (!rows_live(segs, rb * RB, r1)) continue;  // compact rows: none of these 16 exists
    for (long long r = rb * RB; r < r1; ++r) {
      const float* xr = x + r * ldx;
      const float x0 = xr[0], x1 = d > 1 ? xr[1] : 0.f, x2 = d > 2 ? xr[2] : 0.f, x3 = d > 3 ? xr[3] : 0.f;
      // one explicit fma chain starting from v (a multiply followed by an add would be the compiler's to contract or not)
      float o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float acc = __fmaf_rn(x0, w[u].x, u == 0 ? vv.x : u == 1 ? vv.y : u == 2 ? vv.z : vv.w);
        if (d > 1) acc = __fmaf_rn(x1, w[u].y, acc);
        if (d > 2) acc = __fmaf_rn(x2, w[u].z, acc);
        if (d > 3) acc = __fmaf_rn(x3, w[u].w, acc);
        o[u] = acc;
      }
      *reinterpret_cast<float4*>(out + r * ldo + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// shared by the C entry point and the forward driver (which may pass queries folded onto the row input)
static int attn_reduce_impl(const AttnArgs& a_in, int H, hipStream_t st) {
  static const bool span_off = [] { const char* e = getenv("TGMX_ATTN_SPAN"); return e && atoi(e) == 0; }();  // A/B knob
  AttnArgs a = a_in;
  a.full_span = span_off ? 1 : 0;
  TGMX_REQUIRE(H == 1 || H == 2 || H == 4 || H == 8, "tgat_attn_reduce: the attention kernels are built for n_heads in {1, 2, 4, 8} (got %d)", H);
  const int k = a.k, T = a.T;
  const long long R = a.R;
  const size_t per_wave = ((size_t)k * T + (size_t)k * (H + 2)) * sizeof(float);
  int waves = 4;
  while (waves > 1 && per_wave * waves > 64 * 1024) waves >>= 1;
  TGMX_REQUIRE(per_wave * waves <= 64 * 1024, "tgat_attn_reduce: k*T=%d too large for the LDS time-encoding cache", k * T);
  const dim3 grid((unsigned)((R + waves - 1) / waves)), block(waves * kWave);
  const size_t lds = per_wave * waves;
  if ((H == 1 && launch_attn_reg<1>(st, a)) || (H == 2 && launch_attn_reg<2>(st, a))) {
    TGMX_CHECK_LAUNCH("tgat_attn_reduce(reg)");
    return TGMX_OK;
  }
  if (a.qlane) {  // only the register kernel evaluates the folded queries itself
    set_error("tgat_attn_reduce: in-kernel folded queries need the register-resident kernel");
    return TGMX_E_UNSUPPORTED;
  }
  for (int i = 0; i < TGMX_TGAT_MAX_LAYERS; ++i)
    if (a.seg_uniq[i] || a.seg_live[i] || a.seg_nidx[i]) {  // only the register kernel reads rows / neighbor features by index
      set_error("tgat_attn_reduce: compact rows need the register-resident kernel");
      return TGMX_E_UNSUPPORTED;
    }
  for (int i = 0; i < TGMX_TGAT_MAX_LAYERS; ++i)
    if (a.seg_eid[i]) {  // only the register kernel gathers edge features by id
      set_error("tgat_attn_reduce: edge features by id need the register-resident kernel (n_heads <= 2, k <= 20, D %% 4 == 0)");
      return TGMX_E_UNSUPPORTED;
    }
  // slots per score group: the smallest instantiated G >= min(k, 64 / H)
  const int want = k < 64 / H ? k : 64 / H;
#define TGMX_ATTN(H_, G_) launch_attn<H_, G_>(grid, block, lds, st, a)
#define TGMX_ATTN_H(H_)                                     \
  do {                                                      \
    if (want <= 4 && 4 * H_ <= 64) TGMX_ATTN(H_, 4);        \
    else if (want <= 8 && 8 * H_ <= 64) TGMX_ATTN(H_, 8);   \
    else if (want <= 10 && 10 * H_ <= 64) TGMX_ATTN(H_, 10); \
    else if (want <= 16 && 16 * H_ <= 64) TGMX_ATTN(H_, 16); \
    else if (want <= 20 && 20 * H_ <= 64) TGMX_ATTN(H_, 20); \
    else TGMX_ATTN(H_, 64 / H_);                            \
  } while (0)
  switch (H) {
    case 1: TGMX_ATTN_H(1); break;
    case 2: TGMX_ATTN_H(2); break;
    case 4: TGMX_ATTN_H(4); break;
    default: TGMX_ATTN_H(8); break;
  }
#undef TGMX_ATTN_H
#undef TGMX_ATTN
  TGMX_CHECK_LAUNCH("tgat_attn_reduce");
  return TGMX_OK;
}

extern "C" int tgmx_tgat_attn_reduce(const float* qf, const float* nbrf, int32_t d, const float* ex, int32_t D,
                                     const int64_t* seed_t, const int64_t* nbr_t, const int32_t* nbr_id, const float* tw,
                                     const float* tb, const float* nbr_time_feat, const uint8_t* mask, int32_t T, int32_t H,
                                     int32_t k, int64_t R, float scale, int32_t head_stride, float* zbar, float* attn_probs,
                                     const tgmx_dropout_t* drop, tgmx_stream_t stream) {
  TGMX_REQUIRE(d > 0 && D >= 0 && T > 0 && k > 0 && R >= 0, "tgat_attn_reduce: bad sizes d=%d D=%d T=%d k=%d", d, D, T, k);
  TGMX_REQUIRE(H == 1 || H == 2 || H == 4 || H == 8, "tgat_attn_reduce: n_heads=%d unsupported (1, 2, 4 or 8)", H);
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(qf && nbrf && (D == 0 || ex) && zbar, "tgat_attn_reduce: null pointer");
  TGMX_REQUIRE(nbr_time_feat || (seed_t && nbr_t && tw && tb), "tgat_attn_reduce: need times + Time2Vec params or nbr_time_feat");
  TGMX_REQUIRE(mask || nbr_id, "tgat_attn_reduce: need nbr_id or mask");
  TGMX_REQUIRE(head_stride == 0 || head_stride >= d + D + T, "tgat_attn_reduce: head_stride smaller than d + D + T");
  AttnArgs a{qf, nbrf, ex, seed_t, nbr_t, nbr_id, tw, tb, nbr_time_feat, mask, zbar, R, d, D, T, k, d + D + T, scale,
             head_stride ? head_stride : d + D + T, attn_probs};
  a.drop = make_dropout(drop);
  a.drop_row0 = drop ? drop->row0 : 0;
  return attn_reduce_impl(a, H, (hipStream_t)stream);
}

namespace tgmx {
__global__ __launch_bounds__(256) void dropout_kernel(const float* x, long long ldx, long long R, int C, DropoutArgs d, long long row0,
                                                      float* out, long long ldo) {
  const long long total = R * C;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / C;
    const int c = (int)(e - r * C);
    out[r * ldo + c] = x[r * ldx + c] * dropout_scale(d, (unsigned long long)(row0 + r) * C + c);
  }
}
}  // namespace tgmx

extern "C" int tgmx_dropout(const float* x, int64_t ldx, int64_t R, int32_t C, const tgmx_dropout_t* drop, float* out, int64_t ldo,
                            tgmx_stream_t stream) {
  TGMX_REQUIRE(R >= 0 && C > 0 && ldx >= C && ldo >= C, "dropout: bad sizes R=%lld C=%d", (long long)R, C);
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(x && out, "dropout: null pointer");
  TGMX_REQUIRE(!drop || (drop->p >= 0.f && drop->p < 1.f), "dropout: p must be in [0, 1)");
  long long blocks = (R * C + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, (long long)R, C, make_dropout(drop),
                     drop ? (long long)drop->row0 : 0ll, out, (long long)ldo);
  TGMX_CHECK_LAUNCH("dropout");
  return TGMX_OK;
}

extern "C" int tgmx_time2vec(const void* x, int32_t x_is_int64, const float* w, const float* b, int32_t T, int64_t n,
                             float* out, tgmx_stream_t stream) {
  TGMX_REQUIRE(T > 0 && n >= 0, "time2vec: bad sizes");
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(x && w && b && out, "time2vec: null pointer");
  long long blocks = (n * T + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (x_is_int64)
    hipLaunchKernelGGL(time2vec_kernel<int64_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const int64_t*)x, w, b,
                       T, (long long)n, out);
  else
    hipLaunchKernelGGL(time2vec_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, b, T,
                       (long long)n, out);
  TGMX_CHECK_LAUNCH("time2vec");
  return TGMX_OK;
}

// ---------------------------------------------------------------------------
// Whole TGAT forward as ONE host call (tgm/nn/encoder/tgat.py:95-149): leaf gathers, then per
// layer the row-batched sequence rres -> Q -> folded query -> per-level attention -> W_V fold ->
// W_O -> LayerNorm+concat -> merge MLP.  ~25 launches enqueued back to back from C++: the Python
// side only fills two small structs.  Workspace layout is private to this function.
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t x) { return (x + 63) & ~(size_t)63; }
static inline int pad4(int x) { return (x + 3) & ~3; }

extern "C" size_t tgmx_pair_dedup_workspace_bytes(int64_t n);
extern "C" int tgmx_pair_dedup(const int32_t* ids, const int64_t* times, int64_t n, int32_t* uniq, int32_t* owner, int32_t* cidx, int32_t* count,
                               void* workspace, size_t workspace_bytes, tgmx_stream_t stream);

// Inference over the DISTINCT rows of every level (tgmx_tgat_hop_t.seed_keyed): at least two layers (a one-layer model has no level that
// is both a row batch and somebody's neighbors), every hop >= 1 marked, and every layer's attention on the register-resident kernel
// (the only one that reads rows and neighbor features by index).
// OPT-IN -- the caller marks the hops (tgm_amd.nn.TGAT(...).compact_rows = True, or TGMX_TGAT_COMPACT=1 in its environment) -- since it was measured (round 4, headline shape, edge features by id, one box, us per forward):
//   row per slot 192.3 | compact rows 201.9 (pair table in global memory) / 205.0 (one workgroup, table in LDS).
// 12 000 level-1 rows hold ~6 300 distinct pairs, but (i) finding them costs 19-21 us (a memset + two launches of dependent global
// atomics, or one workgroup's serialized LDS atomics) + 5 us for the row maps against 4.8 us for the plain leaf gather, (ii) the one-kernel
// tail is ONE round of 197 workgroups on 256 CUs whose duration is a single workgroup's latency (58.5 us with 109 live workgroups as with
// 197), (iii) the attention launch gains 4-10 us (58 -> 48-54: the all-pad rows it drops were cheap already) and the 600-row layer's
// attention LOSES 2-3 us reading its neighbors through the row map.  It pays from ~2 rounds of tail workgroups on (>= 33 k rows per layer).
static bool compact_wanted(const tgmx_tgat_model_t* m, const tgmx_tgat_hop_t* hops, int save) {
  static const bool off = [] { const char* e = getenv("TGMX_TGAT_COMPACT"); return e && atoi(e) == 0; }();  // kill switch
  const int L = m->num_layers;
  if (off || save || L < 2) return false;
  for (int i = 1; i < L; ++i)
    if (!hops[i].seed_keyed) return false;
  for (int j = 0; j < L; ++j) {
    const tgmx_tgat_layer_t& ly = m->layers[j];
    if (!(ly.H <= 2 && hops[0].k <= 20 && ly.D > 0 && ly.D % 4 == 0 && ly.D / 4 <= 64 && ly.T <= 128 && ((ly.d % 4 == 0 && ly.d / 4 <= 64) || ly.d <= 64))) return false;
    if (!ly.qf_U) return false;  // (the folded query side: the inference path proper)
  }
  for (int i = 0; i < L; ++i) {
    if (hops[i].k != hops[0].k) return false;
    if (!hops[i].edge_x && !(hops[i].nbr_eid && hops[i].edge_table)) return false;
    if (((uintptr_t)hops[i].edge_x & 15) || ((uintptr_t)hops[i].edge_table & 15)) return false;
  }
  return true;
}

// Workspace layout of tgmx_tgat_forward (offsets in floats from the 256-byte aligned base).  Internal
// activation rows are padded so that every GEMM operand row starts 16-byte aligned: O -> Op, per-head
// dh -> dhp, C -> Cp, O + d0 -> Kc, emb -> Ep.  With save == 0 the per-layer scratch is reused by every
// layer; with save != 0 every layer keeps its own (the backward pass reads them).
extern "C" int tgmx_tgat_layout(const tgmx_tgat_model_t* m, int64_t S0, const tgmx_tgat_hop_t* hops, int32_t save,
                                tgmx_tgat_layout_t* out) {
  TGMX_REQUIRE(m && hops && out, "tgat_layout: null pointer");
  const int L = m->num_layers;
  TGMX_REQUIRE(L >= 1 && L <= TGMX_TGAT_MAX_LAYERS, "tgat_layout: num_layers=%d outside [1, %d]", L, TGMX_TGAT_MAX_LAYERS);
  long long rows[TGMX_TGAT_MAX_LAYERS + 1];
  rows[0] = S0;
  for (int i = 0; i < L; ++i) rows[i + 1] = rows[i] * hops[i].k;
  out->level_off[0] = 0;
  for (int i = 0; i <= L; ++i) {
    out->level_rows[i] = rows[i];
    out->level_off[i + 1] = out->level_off[i] + rows[i];
  }
  size_t w = 0;
  auto take = [&](size_t n) { const size_t p = w; w += align_up(n); return (int64_t)p; };
  out->z0 = take((size_t)out->level_off[L + 1] * m->d0);
  size_t scratch0 = w, scratch_end = w;
  for (int j = 1; j <= L; ++j) {
    const tgmx_tgat_layer_t& ly = m->layers[j - 1];
    tgmx_tgat_layer_layout_t& lo = out->layers[j - 1];
    const size_t R = (size_t)out->level_off[L - j + 1];
    const int H = ly.H, k = hops[j - 1].k;
    lo.R = (int64_t)R;
    lo.Op = pad4(ly.O); lo.dhp = pad4(ly.O / H); lo.Cp = pad4(ly.d + ly.D + ly.T); lo.Kc = pad4(ly.O + m->d0); lo.Ep = pad4(ly.emb);
    if (!save) w = scratch0;
    lo.rres = take(R * lo.Op);
    lo.oattn = take(R * lo.Op);
    lo.y = take(R * lo.Op);
    lo.Q = take(R * H * lo.dhp);
    lo.qf = take(R * H * lo.Cp);
    lo.zbar = take(R * H * lo.Cp);
    lo.cat = take(R * lo.Kc);
    lo.h1 = take(R * lo.Ep);
    lo.probs = save ? take(R * H * k) : -1;
    if (save) lo.out = (j == L) ? -1 : take(R * ly.emb_out);
    scratch_end = w > scratch_end ? w : scratch_end;
  }
  w = scratch_end;
  if (!save) {  // ping-pong buffers for the layer outputs
    size_t emb_rows = 0;
    for (int j = 1; j < L; ++j) {
      const size_t e = (size_t)out->level_off[L - j + 1] * m->layers[j - 1].emb_out;
      emb_rows = e > emb_rows ? e : emb_rows;
    }
    const int64_t pp[2] = {take(emb_rows), take(emb_rows)};
    for (int j = 1; j <= L; ++j) out->layers[j - 1].out = (j == L) ? -1 : pp[j & 1];
  }
  // compact rows (tgmx_tgat_hop_t.seed_keyed): one block of int32 per deduplicated level
  for (int i = 0; i <= TGMX_TGAT_MAX_LAYERS; ++i) out->compact[i] = -1;
  if (compact_wanted(m, hops, save)) {
    for (int i = 1; i < L; ++i) {
      const size_t n = (size_t)rows[i];
      out->compact[i] = take(16 + 4 * n + (tgmx_pair_dedup_workspace_bytes((int64_t)n) + 3) / 4);
    }
  }
  out->total_bytes = (int64_t)(w * sizeof(float) + 256);
  return TGMX_OK;
}

extern "C" size_t tgmx_tgat_workspace_bytes(const tgmx_tgat_model_t* m, int64_t S0, const tgmx_tgat_hop_t* hops) {
  tgmx_tgat_layout_t lay;
  return tgmx_tgat_layout(m, S0, hops, 0, &lay) == TGMX_OK ? (size_t)lay.total_bytes : 0;
}

static int launch_post_chain(const tgmx_tgat_layer_t& ly, const tgmx_tgat_layer_layout_t& lo, const float* zbar, const float* x,
                             long long ldx, const float* tb, const float* z0, int d0, long long R, float* out, long long ldo,
                             hipStream_t st, const RowSegs& segs) {
  ChainArgs g{};
  g.segs = segs;
  g.zbar = zbar; g.x = x; g.tb = tb; g.z0 = z0;
  g.W_V = ly.W_V; g.W_O = ly.W_O; g.b_O = ly.b_O; g.ln_g = ly.ln_g; g.ln_b = ly.ln_b;
  g.fc1_w = ly.fc1_w; g.fc1_b = ly.fc1_b; g.fc2_w = ly.fc2_w; g.fc2_b = ly.fc2_b;
  g.out = out; g.R = R; g.ld_zbar = (long long)ly.H * lo.Cp; g.ldx = ldx; g.ldo = ldo;
  g.d = ly.d; g.T = ly.T; g.d0 = d0; g.O = ly.O; g.H = ly.H; g.C = ly.d + ly.D + ly.T; g.emb = ly.emb; g.emb_out = ly.emb_out;
  g.Cp = lo.Cp; g.Op = lo.Op; g.Kc = lo.Kc; g.Ep = lo.Ep; g.eps = ly.ln_eps;
  int kmax = ly.O + d0;
  if (ly.emb > kmax) kmax = ly.emb;
  g.LD = (kmax + 31) / 32 * 32 + 4;
  const size_t lds = ((size_t)64 * g.LD + ly.T + 2 * ly.O + 32 * ly.d + 32 * d0) * sizeof(float);
  if (lds > 160 * 1024) {
    set_error("tgat_forward: layer too wide for the fused chain (%zu bytes of LDS)", lds);
    return TGMX_E_UNSUPPORTED;
  }
  static size_t lds_limit = 0;
  if (lds > lds_limit) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tgat_post_chain_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("tgat_forward: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return TGMX_E_LAUNCH;
    }
    lds_limit = lds;
  }
  hipLaunchKernelGGL(tgat_post_chain_kernel, dim3((unsigned)((R + 31) / 32)), dim3(kChainThreads), lds, st, g);
  TGMX_CHECK_LAUNCH("tgat_post_chain");
  return TGMX_OK;
}

// row-major weight [N, K] (row stride ldw) -> the 16 x 16 tiles chain64_gemm streams, k-block major: [KB][NB][256]
__global__ __launch_bounds__(256) void tile16_kernel(const float* __restrict__ W, long long ldw, int N, int K, float* __restrict__ out) {
  const int KB = (K + 15) / 16;
  const long long total = (long long)((N + 15) / 16) * KB * 256;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long tile = e >> 8;
    const int NBk = (N + 15) / 16;
    const int within = (int)(e & 255), ln = within >> 2, j = within & 3;
    const int n = (int)(tile % NBk) * 16 + (ln & 15), k = (int)(tile / NBk) * 16 + 4 * (ln >> 4) + j;
    out[e] = (n < N && k < K) ? W[(long long)n * ldw + k] : 0.f;
  }
}

struct PackJobs {
  tgmx_pack_job_t job[TGMX_PACK_MAX_JOBS];
};
// blockIdx.y = job; the job's dst elements strided over blockIdx.x
__global__ __launch_bounds__(256) void pack2d_kernel(const PackJobs J) {
  const tgmx_pack_job_t j = J.job[blockIdx.y];
  const long long total = (long long)j.dst_rows * j.dst_cols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(e / j.dst_cols), c = (int)(e - (long long)r * j.dst_cols);
    float v = 0.f;
    if (r < j.rows && c < j.cols) v = j.transpose ? j.src[(long long)c * j.src_ld + r] : j.src[(long long)r * j.src_ld + c];
    j.dst[(long long)r * j.dst_ld + c] = v;
  }
}

extern "C" int tgmx_pack2d(const tgmx_pack_job_t* jobs, int32_t n_jobs, tgmx_stream_t stream) {
  TGMX_REQUIRE(n_jobs >= 0 && n_jobs <= TGMX_PACK_MAX_JOBS, "pack2d: %d jobs (at most %d per call)", n_jobs, TGMX_PACK_MAX_JOBS);
  if (n_jobs == 0) return TGMX_OK;
  TGMX_REQUIRE(jobs, "pack2d: null pointer");
  PackJobs J{};
  long long most = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const tgmx_pack_job_t& j = jobs[i];
    TGMX_REQUIRE(j.src && j.dst && j.rows >= 0 && j.cols >= 0 && j.dst_rows >= j.rows && j.dst_cols >= j.cols && j.dst_ld >= j.dst_cols &&
                     j.src_ld >= (j.transpose ? j.rows : j.cols),
                 "pack2d: job %d is malformed", i);
    J.job[i] = j;
    const long long total = (long long)j.dst_rows * j.dst_cols;
    most = total > most ? total : most;
  }
  if (most == 0) return TGMX_OK;
  long long bx = (most + 1023) / 1024;  // ~4 elements per thread
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(pack2d_kernel, dim3((unsigned)bx, (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, J);
  TGMX_CHECK_LAUNCH("pack2d");
  return TGMX_OK;
}

extern "C" size_t tgmx_tgat_tile16_floats(int32_t N, int32_t K) {
  return N > 0 && K > 0 ? (size_t)((N + 15) / 16) * (size_t)((K + 15) / 16) * 256 : 0;
}

extern "C" int tgmx_tgat_tile16(const float* W, int64_t ldw, int32_t N, int32_t K, float* out, tgmx_stream_t stream) {
  TGMX_REQUIRE(N >= 0 && K >= 0 && ldw >= K, "tgat_tile16: bad shape");
  if (N == 0 || K == 0) return TGMX_OK;
  TGMX_REQUIRE(W && out, "tgat_tile16: null pointer");
  const long long total = (long long)tgmx_tgat_tile16_floats(N, K);
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(tile16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, (long long)ldw, N, K, out);
  TGMX_CHECK_LAUNCH("tgat_tile16");
  return TGMX_OK;
}

// The transposed 16-row-tile chain (tgat_chain64_kernel).  Returns TGMX_E_UNSUPPORTED (no error text) when the layer does not
// fit it (a stage wider than 192 columns, no tiled weights) -- the caller falls back to tgat_post_chain_kernel / the unfused kernels.
static int launch_chain64(const tgmx_tgat_layer_t& ly, const tgmx_tgat_layer_layout_t& lo, const float* zbar, const float* x,
                          long long ldx, const float* tb, const float* z0, int d0, long long R, float* out, long long ldo,
                          hipStream_t st, bool dry, const RowSegs& segs) {
  static const bool off = [] { const char* e = getenv("TGMX_CHAIN64"); return e && atoi(e) == 0; }();  // A/B knob
  if (off || !ly.W_V_t16 || !ly.W_O_t16 || !ly.fc1_t16 || !ly.fc2_t16) return TGMX_E_UNSUPPORTED;
  const int blocks[4] = {(ly.O / ly.H + 15) / 16, (ly.O + 15) / 16, (ly.emb + 15) / 16, (ly.emb_out + 15) / 16};
  for (int b : blocks)
    if (b > kC64MaxB) return TGMX_E_UNSUPPORTED;
  ChainArgs g{};
  g.segs = segs;
  g.zbar = zbar; g.x = x; g.tb = tb; g.z0 = z0;
  g.W_V = ly.W_V_t16; g.W_O = ly.W_O_t16; g.b_O = ly.b_O; g.ln_g = ly.ln_g; g.ln_b = ly.ln_b;
  static const bool both_knob = [] { const char* e = getenv("TGMX_C64_BOTH_HEADS"); return !(e && e[0] == '0'); }();  // A/B knob
  g.W_Vc = (both_knob && ly.H == 2 && 2 * blocks[0] <= kC64MaxB) ? ly.W_V_t16c : nullptr;
  g.fc1_w = ly.fc1_t16; g.fc1_b = ly.fc1_b; g.fc2_w = ly.fc2_t16; g.fc2_b = ly.fc2_b;
  g.out = out; g.R = R; g.ld_zbar = (long long)ly.H * lo.Cp; g.ldx = ldx; g.ldo = ldo;
  g.d = ly.d; g.T = ly.T; g.d0 = d0; g.O = ly.O; g.H = ly.H; g.C = ly.d + ly.D + ly.T; g.emb = ly.emb; g.emb_out = ly.emb_out;
  g.Cp = lo.Cp; g.Op = lo.Op; g.Kc = lo.Kc; g.Ep = lo.Ep; g.eps = ly.ln_eps;
  // slab row: every stage's padded width (stage 1 writes 16-column blocks from column (H - 1) dh on)
  int kmax = ly.O + d0;
  if (ly.emb > kmax) kmax = ly.emb;
  if ((ly.H - 1) * (ly.O / ly.H) + 16 * blocks[0] > kmax) kmax = (ly.H - 1) * (ly.O / ly.H) + 16 * blocks[0];
  g.LD = (kmax + 15) / 16 * 16 + 4;
  const size_t lds = ((size_t)kC64Ring * kC64ChunkTiles * 256 + kC64Waves * 256 + (size_t)kC64Tiles * 16 * g.LD + 16 * (size_t)(3 * blocks[1] + blocks[2] + blocks[3]) + kC64Waves * 16 +
                      64 * (size_t)ly.d + 64 * (size_t)d0) * sizeof(float);
  if (lds > 160 * 1024) return TGMX_E_UNSUPPORTED;
  if (dry) return TGMX_OK;
  // the opt-in to > 64 KB of dynamic LDS is a per-DEVICE function attribute: one flag per device id, set with a release store so that
  // two threads racing on it at worst set the attribute twice (idempotent)
  constexpr int kMaxDevices = 64;
  static std::atomic<bool> attr_set[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
  if (dev < 0 || !attr_set[dev].load(std::memory_order_acquire)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tgat_chain64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("tgat_forward: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return TGMX_E_LAUNCH;
    }
    if (dev >= 0) attr_set[dev].store(true, std::memory_order_release);
  }
  hipLaunchKernelGGL(tgat_chain64_kernel, dim3((unsigned)((R + 63) / 64)), dim3(kC64Threads), lds, st, g);
  TGMX_CHECK_LAUNCH("tgat_chain64");
  return TGMX_OK;
}

extern "C" int tgmx_tgat_forward(const tgmx_tgat_model_t* m, const float* node_x, int64_t num_nodes, const int32_t* seed_ids,
                                 int64_t S0, const tgmx_tgat_hop_t* hops, float* workspace, size_t workspace_bytes, int32_t save,
                                 float* out, tgmx_stream_t stream) {
  TGMX_REQUIRE(m && hops && node_x && seed_ids && out, "tgat_forward: null pointer");
  if (S0 == 0) return TGMX_OK;
  tgmx_tgat_layout_t lay;
  int rc = tgmx_tgat_layout(m, S0, hops, save, &lay);
  if (rc) return rc;
  TGMX_REQUIRE(workspace && workspace_bytes >= (size_t)lay.total_bytes, "tgat_forward: workspace too small (%zu < %lld)", workspace_bytes,
               (long long)lay.total_bytes);
  const int L = m->num_layers, d0 = m->d0;
  const int64_t* off = lay.level_off;
  const int64_t* rows = lay.level_rows;
  float* base = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  float* z0 = base + lay.z0;

  // Compact rows (tgmx_tgat_hop_t.seed_keyed): the distinct (id, time) pairs of every level that is a row batch, then the leaf
  // features of the distinct rows only; the deepest level is read from node_x by node id where the attention consumes it.
  const bool compact = lay.compact[1] >= 0;
  struct CompactLevel {
    int32_t *count, *uniq, *owner, *cidx, *rep;
  } cl[TGMX_TGAT_MAX_LAYERS + 1] = {};
  if (compact) {
    CompactGatherArgs cg{};
    cg.levels = L;
    cg.table = node_x; cg.out = z0; cg.num_nodes = num_nodes; cg.dim = d0;
    cg.ids[0] = seed_ids; cg.rows[0] = rows[0]; cg.off[0] = 0;
    cg.itemsB = rows[0] * d0;
    for (int i = 1; i < L; ++i) {
      TGMX_REQUIRE(hops[i - 1].nbr_id && hops[i - 1].nbr_t, "tgat_forward: hop %d has no neighbor ids / times", i - 1);
      const long long n = rows[i];
      int32_t* blk = reinterpret_cast<int32_t*>(base + lay.compact[i]);
      cl[i] = CompactLevel{blk, blk + 16, blk + 16 + n, blk + 16 + 2 * n, blk + 16 + 3 * n};
      if ((rc = tgmx_pair_dedup(hops[i - 1].nbr_id, hops[i - 1].nbr_t, n, cl[i].uniq, cl[i].owner, cl[i].cidx, cl[i].count, blk + 16 + 4 * n,
                                tgmx_pair_dedup_workspace_bytes(n), stream)))
        return rc;
      cg.ids[i] = hops[i - 1].nbr_id; cg.rows[i] = n; cg.off[i] = off[i];
      cg.uniq[i] = cl[i].uniq; cg.owner[i] = cl[i].owner; cg.cidx[i] = cl[i].cidx; cg.rep[i] = cl[i].rep; cg.count[i] = cl[i].count;
      cg.itemsA += n;
      cg.itemsB += n * d0;
    }
    long long blocks = (cg.itemsA + cg.itemsB + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks > 0) hipLaunchKernelGGL(gather_compact_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, cg);
    TGMX_CHECK_LAUNCH("tgat_gather_compact");
  } else {
    LeafGatherArgs lg{};
    lg.idx[0] = seed_ids;
    lg.end[0] = off[1];
    for (int i = 1; i <= L; ++i) {
      TGMX_REQUIRE(rows[i] == 0 || hops[i - 1].nbr_id, "tgat_forward: hop %d has no neighbor ids", i - 1);
      lg.idx[i] = hops[i - 1].nbr_id;
      lg.end[i] = off[i + 1];
    }
    lg.table = node_x; lg.out = z0; lg.rows = num_nodes; lg.levels = L + 1; lg.dim = d0;
    const long long total = off[L + 1] * d0;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (total > 0) hipLaunchKernelGGL(gather_leaves_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lg);
  }

  const float* prev = z0;
  long long ld_prev = d0;
  for (int j = 1; j <= L; ++j) {
    const tgmx_tgat_layer_t& ly = m->layers[j - 1];
    const tgmx_tgat_layer_layout_t& lo = lay.layers[j - 1];
    const int n_lvl = L - j + 1;
    const long long R = lo.R;
    const int O = ly.O, H = ly.H, dh = O / H, C = ly.d + ly.D + ly.T, k = hops[j - 1].k;
    const int Op = lo.Op, dhp = lo.dhp, Cp = lo.Cp, Kc = lo.Kc, Ep = lo.Ep;
    TGMX_REQUIRE(ly.d == (j == 1 ? d0 : m->layers[j - 2].emb_out), "tgat_forward: layer %d input width mismatch", j);
    float *rres = base + lo.rres, *oattn = base + lo.oattn, *y = base + lo.y, *Q = base + lo.Q, *qf = base + lo.qf;
    float *zbar = base + lo.zbar, *cat = base + lo.cat, *h1 = base + lo.h1;
    float* probs = lo.probs >= 0 ? base + lo.probs : nullptr;
    float* nxt = (j == L) ? out : base + lo.out;
    const long long ld_nxt = ly.emb_out;  // layer outputs stay densely packed: they are the next layer's neighbor features
    // inference, enough row tiles to fill the chip: the whole tail of the layer is one kernel over row tiles (both variants
    // rebuild the residual themselves) -- 64-row workgroups of 16-row MFMA tiles when the layer fits that kernel, else 32-row tiles
    const bool chain = !save && R >= 2048;
    const bool chain64 = chain && launch_chain64(ly, lo, nullptr, nullptr, 0, nullptr, nullptr, d0, R, nullptr, 0, nullptr, true, RowSegs{}) == TGMX_OK;
    const bool folded = !save && ly.qf_U != nullptr;  // inference: qf = x . U^T + v in one GEMM, no Q / rres round trip
    RowSegs segs{};  // which rows of this layer's row space exist (compact rows; all of them otherwise)
    if (compact) {
      segs.n = n_lvl;
      for (int i = 0; i < n_lvl; ++i) {
        segs.begin[i] = off[i];
        segs.live[i] = i >= 1 ? cl[i].count : nullptr;
      }
      segs.begin[n_lvl] = off[n_lvl];
    }
    const int dp = (ly.d + 3) / 4 * 4;
    // one scalar of input per row and the register-resident attention kernel takes the layer (launch_attn_reg's conditions): the
    // folded queries are evaluated inside that kernel (TGMX_TGAT_QF_INLINE=0: the A/B knob)
    static const bool qf_knob = [] { const char* e = getenv("TGMX_TGAT_QF_INLINE"); return !(e && e[0] == '0'); }();
    bool qf_in_kernel = folded && qf_knob && ly.qf_lane && ly.d == 1 && ld_prev == 1 && H <= 2 && k <= 20 && ly.D > 0 && ly.D % 4 == 0 &&
                        ly.D / 4 <= 64 && ly.T <= 128;
    for (int i = 0; i < n_lvl && qf_in_kernel; ++i)
      qf_in_kernel = hops[i].edge_x ? ((uintptr_t)hops[i].edge_x & 15) == 0 : (hops[i].nbr_eid && ((uintptr_t)hops[i].edge_table & 15) == 0);
    if (!folded)  // the residual as a buffer is the Q projection's input; LayerNorm rebuilds it on the fly
      if ((rc = tgmx_tgat_rres(prev, ld_prev, ly.d, m->tb, nullptr, ly.T, O, R, rres, Op, stream))) return rc;
    if (!folded) {
      // Q[:, head h] = rres @ W_Q[head h rows]^T     (heads written dhp apart)
      if ((rc = tgmx_sgemm_nt(rres, Op, ly.W_Q, Op, Q, (long long)H * dhp, R, dh, O, nullptr, 0, H, 0, (long long)dh * Op, dhp, stream))) return rc;
      // qf[:, h, :] = Q[:, head h] @ W_K[head h]  via the transposed padded copy W_K_t [C, H*dhp]
      if ((rc = tgmx_sgemm_nt(Q, (long long)H * dhp, ly.W_K_t, (long long)H * dhp, qf, (long long)H * Cp, R, C, dh, nullptr, 0, H, dhp, dhp, Cp, stream)))
        return rc;
    } else if (qf_in_kernel) {
      // the attention kernel evaluates its query columns itself (AttnArgs.qlane): no qf buffer, no launch
    } else if (ly.d <= 4) {
      long long blocks = ((R + 15) / 16 * (H * Cp / 4) + 255) / 256;
      if (blocks > 16384) blocks = 16384;
      hipLaunchKernelGGL(tgat_qfold_small_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, prev, ld_prev, ly.d, ly.qf_U, ly.qf_v,
                         H * Cp / 4, R, qf, (long long)H * Cp, segs);
      TGMX_CHECK_LAUNCH("tgat_qfold_small");
    } else {
      if ((rc = tgmx_sgemm_nt(prev, ld_prev, ly.qf_U, dp, qf, (long long)H * Cp, R, H * Cp, ly.d, ly.qf_v, 0, 1, 0, 0, 0, stream))) return rc;
    }
    {
      // every level this layer aggregates, in ONE launch (same weights, same k; qf / zbar / probs are contiguous)
      AttnArgs a{};
      bool any_by_id = false;
      a.qf = qf; a.zbar = zbar; a.probs = probs; a.R = off[n_lvl];
      if (qf_in_kernel) { a.qlane = ly.qf_lane; a.qx = prev; }
      a.tw = m->tw; a.tb = m->tb;
      a.d = ly.d; a.D = ly.D; a.T = ly.T; a.k = k; a.C = C; a.scale = 1.0f / sqrtf((float)dh); a.Cs = Cp;
      a.n_seg = n_lvl;
      if (save && m->drop.p > 0.f) a.drop = make_dropout(m->drop.p, m->drop.seed, m->drop.stream * 64 + 2 * (unsigned long long)j);
      for (int i = 0; i < n_lvl; ++i) {
        TGMX_REQUIRE(hops[i].k == k, "tgat_forward: layer %d needs the same k at every hop it aggregates", j);
        TGMX_REQUIRE(rows[i] == 0 || ly.D == 0 || hops[i].edge_x || (hops[i].nbr_eid && hops[i].edge_table), "tgat_forward: hop %d has no edge features", i);
        const bool by_id = !hops[i].edge_x && hops[i].nbr_eid && ly.D > 0;
        const bool reg_kernel = H <= 2 && k <= 20 && ly.D % 4 == 0 && ly.D / 4 <= 64 && ly.T <= 128 && ((ly.d % 4 == 0 && ly.d / 4 <= 64) || ly.d <= 64) &&
                                ((uintptr_t)hops[i].edge_table & 15) == 0;  // launch_attn_reg's own conditions
        if (by_id && !reg_kernel) {  // (round 3, later: the saving forward takes ids too -- tgmx_tgat_backward reads the same rows)
          set_error("tgat_forward: edge features by id need the register-resident attention kernel");
          return TGMX_E_UNSUPPORTED;
        }
        any_by_id |= by_id;
        (void)any_by_id;
        const long long b = off[i];  // the kernel indexes with the global row: bias every level's arrays by its first row
        a.seg_begin[i] = b;
        a.seg_nbrf[i] = prev + off[i + 1] * ld_prev - b * (long long)k * ly.d;
        a.seg_ex[i] = hops[i].edge_x ? hops[i].edge_x - b * (long long)k * ly.D : nullptr;
        a.seg_eid[i] = by_id ? hops[i].nbr_eid - b * (long long)k : nullptr;
        a.seg_table[i] = by_id ? hops[i].edge_table : nullptr;
        a.seg_seed_t[i] = hops[i].seed_t - b;
        a.seg_nbr_t[i] = hops[i].nbr_t - b * (long long)k;
        a.seg_nbr_id[i] = hops[i].nbr_id - b * (long long)k;
        if (compact) {
          // rows: level i >= 1 holds its distinct rows only.  Neighbor features: the compact rows of level i + 1 through its rep map
          // (the previous layer's outputs, or in layer 1 the compact leaf features), or -- layer 1, deepest level -- node_x by node id
          a.seg_uniq[i] = i >= 1 ? cl[i].uniq - b : nullptr;
          a.seg_live[i] = i >= 1 ? cl[i].count : nullptr;
          a.seg_nbrf[i] = nullptr;
          if (i + 1 < L) {
            a.seg_nidx[i] = cl[i + 1].rep - b * (long long)k;
            a.seg_ntab[i] = prev + off[i + 1] * ld_prev;
            a.seg_npad[i] = 0;  // (a rep map has no negative entries)
          } else {
            a.seg_nidx[i] = hops[i].nbr_id - b * (long long)k;
            a.seg_ntab[i] = node_x;
            a.seg_npad[i] = num_nodes - 1;  // pad id -1 reads the LAST row of node_x (tgat.py:128-130)
          }
        }
      }
      a.nbrf = a.seg_nbrf[0]; a.ex = a.seg_ex[0]; a.seed_t = a.seg_seed_t[0]; a.nbr_t = a.seg_nbr_t[0]; a.nbr_id = a.seg_nbr_id[0];
      TGMX_REQUIRE(ld_prev == ly.d, "tgat_forward: layer %d expects densely packed input rows", j);
      if (a.R > 0 && (rc = attn_reduce_impl(a, H, (hipStream_t)stream))) return rc;
    }
    if (chain64) {
      if ((rc = launch_chain64(ly, lo, zbar, prev, ld_prev, m->tb, z0, d0, R, nxt, ld_nxt, (hipStream_t)stream, false, segs))) return rc;
      prev = nxt;
      ld_prev = ld_nxt;
      continue;
    }
    if (chain) {  // intermediates stay in LDS
      if ((rc = launch_post_chain(ly, lo, zbar, prev, ld_prev, m->tb, z0, d0, R, nxt, ld_nxt, (hipStream_t)stream, segs))) return rc;
      prev = nxt;
      ld_prev = ld_nxt;
      continue;
    }
    // Oattn[:, head h] = zbar[:, h, :] @ W_V[head h]^T   (W_V padded copy [O, Cp])
    if ((rc = tgmx_sgemm_nt(zbar, (long long)H * Cp, ly.W_V, Cp, oattn, Op, R, dh, C, nullptr, 0, H, Cp, (long long)dh * Cp, dh, stream))) return rc;
    if ((rc = tgmx_sgemm_nt(oattn, Op, ly.W_O, Op, y, Op, R, O, O, ly.b_O, 0, 1, 0, 0, 0, stream))) return rc;
    if (save && m->drop.p > 0.f) {  // dropout(W_O(O)) (attention.py:126), in place: the saved y is what the LayerNorm saw
      const tgmx_dropout_t dy{m->drop.p, m->drop.seed, m->drop.stream * 64 + 2 * (uint64_t)j + 1, 0};
      if ((rc = tgmx_dropout(y, Op, R, O, &dy, y, Op, stream))) return rc;
    }
    if ((rc = ln_residual_concat_impl(y, Op, folded ? nullptr : rres, Op, ly.ln_g, ly.ln_b, O, ly.ln_eps, z0, d0, R, cat, Kc, prev, ld_prev, ly.d, m->tb,
                                      ly.T, stream)))
      return rc;
    if ((rc = tgmx_sgemm_nt(cat, Kc, ly.fc1_w, Kc, h1, Ep, R, ly.emb, O + d0, ly.fc1_b, 1, 1, 0, 0, 0, stream))) return rc;
    if ((rc = tgmx_sgemm_nt(h1, Ep, ly.fc2_w, Ep, nxt, ld_nxt, R, ly.emb_out, ly.emb, ly.fc2_b, 0, 1, 0, 0, 0, stream))) return rc;
    prev = nxt;
    ld_prev = ld_nxt;
  }
  return TGMX_OK;
}
