// Static per-node temporal index (CSR) build, fully on the device.
//
// Node n's adjacency entries must be ordered by (batch_idx, time, role, eid), role 0 = n is the edge's source.
// Because the stream is time-sorted and batches are contiguous edge ranges, that order is a closed-form position:
// inside a maximal run [lo, hi) of edges sharing (batch, timestamp) the source-role entries come first, then the
// destination-role entries, so   pos(e, src-role) = lo + e,   pos(e, dst-role) = hi + e   (a permutation of
// [0, 2E)).  One LSD radix sort (rocPRIM) of the 64-bit key (node << 32 | pos) then yields the whole index: keys are
// unique, so no stability argument is needed; indptr is a binary search per node over the sorted keys (no atomics,
// no scan), and the records are packed straight from the sorted values.
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace tgmx {

struct __align__(16) CsrRec {
  int nbr;
  int eid;
  long long ts;
};
static_assert(sizeof(CsrRec) == sizeof(tgmx_adj_t), "record layout");

struct CsrKeyArgs {
  const int32_t* src;
  const int32_t* dst;
  const int64_t* ts;
  const int64_t* starts;  // [nb] first edge of every batch (increasing); edges before starts[0] form a leading batch
  long long E, nb;
  unsigned long long* keys;
  unsigned int* vals;
  int32_t* status;
  int N, directed;
};

__global__ __launch_bounds__(256) void csr_keys_kernel(const CsrKeyArgs a) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.E) return;
  const int s = a.src[e], d = a.dst[e];
  if (s < 0 || s >= a.N || d < 0 || d >= a.N) atomicOr(a.status, TGMX_ST_EDGE_RANGE);
  if (a.directed) {
    a.keys[e] = ((unsigned long long)(unsigned)s << 32) | (unsigned long long)e;
    a.vals[e] = (unsigned)e;
    return;
  }
  long long run_lo = e, run_hi = e + 1;  // nb < 0: every edge is its own batch -> plain (eid, role) order
  if (a.nb >= 0) {
    // batch of e: number of starts <= e
    long long lo = 0, hi = a.nb;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (a.starts[mid] <= e) lo = mid + 1;
      else hi = mid;
    }
    const long long b_lo = lo > 0 ? a.starts[lo - 1] : 0;
    const long long b_hi = lo < a.nb ? a.starts[lo] : a.E;
    // run of equal timestamps around e, clamped to the batch (ts is sorted)
    const long long t = a.ts[e];
    long long l = b_lo, h = e;  // first index in [b_lo, e] with ts == t
    while (l < h) {
      const long long mid = (l + h) >> 1;
      if (a.ts[mid] < t) l = mid + 1;
      else h = mid;
    }
    run_lo = l;
    l = e + 1;
    h = b_hi;  // first index in (e, b_hi] with ts > t
    while (l < h) {
      const long long mid = (l + h) >> 1;
      if (a.ts[mid] <= t) l = mid + 1;
      else h = mid;
    }
    run_hi = l;
  }
  a.keys[e] = ((unsigned long long)(unsigned)s << 32) | (unsigned long long)(run_lo + e);
  a.vals[e] = (unsigned)e;
  a.keys[a.E + e] = ((unsigned long long)(unsigned)d << 32) | (unsigned long long)(run_hi + e);
  a.vals[a.E + e] = (unsigned)(a.E + e);
}

__global__ __launch_bounds__(256) void csr_indptr_kernel(const unsigned long long* __restrict__ keys, long long M, int N,
                                                         int64_t* __restrict__ indptr) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  long long lo = 0, hi = M;  // first sorted position whose node >= n
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((long long)(keys[mid] >> 32) < n) lo = mid + 1;
    else hi = mid;
  }
  indptr[n] = lo;
}

__global__ __launch_bounds__(256) void csr_pack_kernel(const unsigned int* __restrict__ vals, long long M, long long E,
                                                       const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                       const int64_t* __restrict__ ts, CsrRec* __restrict__ adj) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const long long c = vals[p];
  const bool rev = c >= E;
  const long long e = rev ? c - E : c;
  CsrRec r;
  r.nbr = rev ? src[e] : dst[e];
  r.eid = (int)e;
  r.ts = ts[e];
  adj[p] = r;
}

static int key_bits(int N) {
  int b = 1;
  while ((1ll << b) < (long long)N) ++b;
  return 32 + b;
}

struct CsrWorkspace {
  size_t keys_in, keys_out, vals_in, vals_out, temp, temp_bytes, total;
};

static int csr_layout(long long E, int N, int directed, CsrWorkspace& w) {
  const long long M = directed ? E : 2 * E;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  w.keys_in = off; off = up(off + (size_t)M * 8);
  w.keys_out = off; off = up(off + (size_t)M * 8);
  w.vals_in = off; off = up(off + (size_t)M * 4);
  w.vals_out = off; off = up(off + (size_t)M * 4);
  size_t tb = 0;
  const hipError_t err = rocprim::radix_sort_pairs(nullptr, tb, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                   (const unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)M, 0u,
                                                   (unsigned)key_bits(N));
  if (err != hipSuccess) {
    set_error("csr_build: radix sort workspace query failed: %s", hipGetErrorString(err));
    return TGMX_E_LAUNCH;
  }
  w.temp = off;
  w.temp_bytes = tb;
  off = up(off + tb);
  w.total = off + 256;  // slack for aligning the caller's base pointer
  return TGMX_OK;
}

}  // namespace tgmx

using namespace tgmx;

extern "C" size_t tgmx_csr_build_workspace_bytes(int64_t num_edges, int32_t num_nodes, int32_t directed) {
  if (num_edges <= 0 || num_nodes <= 0) return 256;
  CsrWorkspace w;
  if (csr_layout(num_edges, num_nodes, directed, w)) return 0;
  return w.total;
}

extern "C" int tgmx_csr_build(const int32_t* src, const int32_t* dst, const int64_t* ts, int64_t num_edges, int32_t num_nodes,
                              const int64_t* batch_starts, int64_t num_batches, int32_t directed, int64_t* indptr,
                              tgmx_adj_t* adj, void* workspace, size_t workspace_bytes, int32_t* status,
                              tgmx_stream_t stream) {
  TGMX_REQUIRE(num_edges >= 0 && num_nodes > 0 && num_batches >= -1, "csr_build: bad sizes E=%lld N=%d nb=%lld", (long long)num_edges,
               num_nodes, (long long)num_batches);
  TGMX_REQUIRE(indptr && status, "csr_build: null pointer");
  TGMX_REQUIRE(2 * num_edges < 2147483647LL, "csr_build: more than 2^30 edges need 64-bit positions");
  hipStream_t st = (hipStream_t)stream;
  const long long M = directed ? num_edges : 2 * num_edges;
  if (num_edges == 0) {
    (void)hipMemsetAsync(indptr, 0, sizeof(int64_t) * ((size_t)num_nodes + 1), st);
    return TGMX_OK;
  }
  TGMX_REQUIRE(src && dst && ts && adj && workspace, "csr_build: null pointer");
  TGMX_REQUIRE(directed || num_batches <= 0 || batch_starts, "csr_build: null batch_starts");
  TGMX_REQUIRE(((uintptr_t)adj & 15) == 0, "csr_build: adj must be 16-byte aligned");
  CsrWorkspace w;
  const int rc = csr_layout(num_edges, num_nodes, directed, w);
  if (rc) return rc;
  TGMX_REQUIRE(workspace_bytes >= w.total, "csr_build: workspace of %zu bytes, need %zu", workspace_bytes, w.total);
  char* base = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  auto* keys_in = reinterpret_cast<unsigned long long*>(base + w.keys_in);
  auto* keys_out = reinterpret_cast<unsigned long long*>(base + w.keys_out);
  auto* vals_in = reinterpret_cast<unsigned int*>(base + w.vals_in);
  auto* vals_out = reinterpret_cast<unsigned int*>(base + w.vals_out);

  CsrKeyArgs k{src, dst, ts, batch_starts, num_edges, num_batches, keys_in, vals_in, status, num_nodes, directed};
  hipLaunchKernelGGL(csr_keys_kernel, dim3((unsigned)((num_edges + 255) / 256)), dim3(256), 0, st, k);
  size_t tb = w.temp_bytes;
  const hipError_t err = rocprim::radix_sort_pairs(base + w.temp, tb, (const unsigned long long*)keys_in, keys_out,
                                                   (const unsigned int*)vals_in, vals_out, (size_t)M, 0u,
                                                   (unsigned)key_bits(num_nodes), st);
  if (err != hipSuccess) {
    set_error("csr_build: radix sort failed: %s", hipGetErrorString(err));
    return TGMX_E_LAUNCH;
  }
  hipLaunchKernelGGL(csr_indptr_kernel, dim3((unsigned)((num_nodes + 1 + 255) / 256)), dim3(256), 0, st, keys_out, M, num_nodes,
                     indptr);
  hipLaunchKernelGGL(csr_pack_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, vals_out, M, (long long)num_edges, src,
                     dst, ts, reinterpret_cast<CsrRec*>(adj));
  TGMX_CHECK_LAUNCH("csr_build");
  return TGMX_OK;
}

// ---------------------------------------------------------------------------
// Group edges by a small integer key (TransformerConv: incoming edges per target node).  The reference stack gets
// there with a stable 64-bit argsort plus two searchsorted calls (~10 launches); here: iota, ONE stable LSD radix sort
// over the ceil(log2 num_keys) bits that can be set, and a binary search per key for the segment bounds.
// ---------------------------------------------------------------------------
namespace tgmx {

__global__ __launch_bounds__(256) void segsort_keys_kernel(const int64_t* __restrict__ key, long long n, int num_keys,
                                                           unsigned int* __restrict__ k32, int64_t* __restrict__ iota,
                                                           int32_t* status) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long k = key[i];
  if (k < 0 || k >= num_keys) atomicOr(status, TGMX_ST_EDGE_RANGE);
  k32[i] = (unsigned int)(k < 0 ? 0 : (k >= num_keys ? num_keys - 1 : k));
  iota[i] = i;
}

__global__ __launch_bounds__(256) void segsort_bounds_kernel(const unsigned int* __restrict__ sorted, long long n, int num_keys,
                                                             int64_t* __restrict__ seg_lo, int64_t* __restrict__ seg_hi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_keys) return;
  auto lower = [&](unsigned int v) {  // first position with sorted[p] >= v
    long long lo = 0, hi = n;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (sorted[mid] < v) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  seg_lo[k] = lower((unsigned int)k);
  seg_hi[k] = lower((unsigned int)k + 1u);
}

struct SegSortLayout {
  size_t k_in, k_out, v_in, temp, temp_bytes, total;
};
static int segsort_layout(long long n, SegSortLayout& w) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = 0;
  w.k_in = off; off = up(off + (size_t)n * 4);
  w.k_out = off; off = up(off + (size_t)n * 4);
  w.v_in = off; off = up(off + (size_t)n * 8);
  size_t tb = 0;
  if (rocprim::radix_sort_pairs(nullptr, tb, (const unsigned int*)nullptr, (unsigned int*)nullptr, (const int64_t*)nullptr,
                                (int64_t*)nullptr, (size_t)n, 0u, 32u) != hipSuccess)
    return TGMX_E_LAUNCH;
  w.temp = off; w.temp_bytes = tb; off = up(off + tb);
  w.total = off + 256;
  return TGMX_OK;
}

}  // namespace tgmx

namespace tgmx {
// Small inputs (n <= 16384 keys below 2^15: the edge list of one TGN batch) in ONE launch and without any library call:
// one workgroup sorts the packed values (key << 14 | index) -- unique, so ascending order IS the stable order -- with a
// bitonic network in LDS (16 values per thread), then writes the permutation and every key's [lo, hi) range.  The rocPRIM
// path costs the host ~60 us of launches for the same result.
constexpr int kSegSmallMax = 16384;
__global__ __launch_bounds__(1024) void segsort_small_kernel(const int64_t* __restrict__ key, int n, int num_keys, int64_t* __restrict__ order,
                                                             int64_t* __restrict__ seg_lo, int64_t* __restrict__ seg_hi,
                                                             int32_t* __restrict__ status) {
  __shared__ unsigned v[kSegSmallMax];
  const int tid = threadIdx.x;
  int P = 1024;
  while (P < n) P <<= 1;
  bool bad = false;
  for (int i = tid; i < P; i += 1024) {
    unsigned x = 0xffffffffu;
    if (i < n) {
      long long kk = key[i];
      if (kk < 0 || kk >= num_keys) {
        bad = true;
        kk = kk < 0 ? 0 : num_keys - 1;
      }
      x = ((unsigned)kk << 14) | (unsigned)i;
    }
    v[i] = x;
  }
  if (bad) atomicOr(status, TGMX_ST_EDGE_RANGE);
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
        const int p = i | j;
        const unsigned a = v[i], b = v[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          v[i] = b;
          v[p] = a;
        }
      }
      __syncthreads();
    }
  }
  // bounds as searchsorted gives them: a key without entries gets lo = hi = the position where it would be inserted
  for (int i = tid; i < n; i += 1024) {
    const unsigned x = v[i];
    const int kk = (int)(x >> 14);
    order[i] = (long long)(x & 0x3fffu);
    const int prev = i == 0 ? -1 : (int)(v[i - 1] >> 14);
    if (prev != kk) {
      seg_lo[kk] = i;
      for (int g = prev + 1; g < kk; ++g) seg_lo[g] = seg_hi[g] = i;  // the absent keys just below this one
    }
    if (i == n - 1 || (int)(v[i + 1] >> 14) != kk) seg_hi[kk] = i + 1;
  }
  const int last = (int)(v[n - 1] >> 14);
  for (int g = last + 1 + tid; g < num_keys; g += 1024) seg_lo[g] = seg_hi[g] = n;
}
}  // namespace tgmx

extern "C" size_t tgmx_segment_sort_workspace_bytes(int64_t n) {
  tgmx::SegSortLayout w;
  if (n <= 0) return 256;
  return tgmx::segsort_layout(n, w) == TGMX_OK ? w.total : 0;
}

extern "C" int tgmx_segment_sort(const int64_t* key, int64_t n, int32_t num_keys, int64_t* order, int64_t* seg_lo, int64_t* seg_hi,
                                 void* workspace, size_t workspace_bytes, int32_t* status, tgmx_stream_t stream) {
  using namespace tgmx;
  TGMX_REQUIRE(n >= 0 && num_keys > 0, "segment_sort: bad sizes n=%lld num_keys=%d", (long long)n, num_keys);
  TGMX_REQUIRE(seg_lo && seg_hi && status, "segment_sort: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    (void)hipMemsetAsync(seg_lo, 0, (size_t)num_keys * 8, st);
    (void)hipMemsetAsync(seg_hi, 0, (size_t)num_keys * 8, st);
    return TGMX_OK;
  }
  TGMX_REQUIRE(key && order && workspace, "segment_sort: null pointer");
  // measured (TGN batch, ~5 k edges): the bitonic network takes 58 us on the device against ~25 us of rocPRIM kernels, and once
  // the pipeline is GPU-bound that outweighs the ~50 us of host launches it saves: off unless TGMX_SEGSORT_SMALL=1
  static const bool small_on = getenv("TGMX_SEGSORT_SMALL") != nullptr;
  if (small_on && n <= kSegSmallMax && num_keys <= (1 << 15)) {
    hipLaunchKernelGGL(segsort_small_kernel, dim3(1), dim3(1024), 0, st, key, (int)n, num_keys, order, seg_lo, seg_hi, status);
    TGMX_CHECK_LAUNCH("segment_sort");
    return TGMX_OK;
  }
  SegSortLayout w;
  if (segsort_layout(n, w) != TGMX_OK || workspace_bytes < w.total) {
    set_error("segment_sort: workspace too small (%zu bytes)", workspace_bytes);
    return TGMX_E_INVALID;
  }
  char* base = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  auto* k_in = reinterpret_cast<unsigned int*>(base + w.k_in);
  auto* k_out = reinterpret_cast<unsigned int*>(base + w.k_out);
  auto* v_in = reinterpret_cast<int64_t*>(base + w.v_in);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(segsort_keys_kernel, dim3(blocks), dim3(256), 0, st, key, (long long)n, num_keys, k_in, v_in, status);
  unsigned bits = 1;
  while ((1ll << bits) < num_keys) ++bits;
  size_t tb = w.temp_bytes;
  if (rocprim::radix_sort_pairs(base + w.temp, tb, (const unsigned int*)k_in, k_out, (const int64_t*)v_in, order, (size_t)n,
                                0u, bits, st) != hipSuccess) {
    set_error("segment_sort: radix sort failed");
    return TGMX_E_LAUNCH;
  }
  hipLaunchKernelGGL(segsort_bounds_kernel, dim3((unsigned)((num_keys + 255) / 256)), dim3(256), 0, st, k_out, (long long)n, num_keys,
                     seg_lo, seg_hi);
  TGMX_CHECK_LAUNCH("segment_sort");
  return TGMX_OK;
}

