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

#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/block/block_scan.hpp>
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
// Small inputs (n <= 16384 keys below 2^15: the edge list of one TGN batch) in ONE launch, one workgroup, no library call.
// The edge list of a sampled batch is seed-major: the k slots of a seed share a target, so the keys come in RUNS (at most one
// per seed: 1536 for a 512-edge batch with negatives, against ~15 000 edges).  Sorting the run HEADS -- packed (key << 12 | run
// number), unique, so ascending order IS the stable order -- with a bitonic network in LDS is a 2048-element problem; the runs'
// lengths in sorted order are scanned into output offsets and every output position finds its run by binary search.  Keys that do
// not come in runs (more than 4096 of them) take the same network over all elements (16 values per thread, ~58 us), still one
// launch and the same result.  (Round 2 measured only that element-wise network and kept the rocPRIM chain -- keys, six sort
// launches, bounds: 43 us of kernels -- for being faster.  It still is for a TGN batch, whose targets are the neighbours:
// see the knob in tgmx_segment_sort.)
constexpr int kSegSmallMax = 16384, kSegRunsMax = 4096;

__device__ __forceinline__ void lds_bitonic(unsigned* v, int P, int tid) {
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
}

__global__ __launch_bounds__(1024) void segsort_small_kernel(const int64_t* __restrict__ key, int n, int num_keys, int64_t* __restrict__ order,
                                                             int64_t* __restrict__ seg_lo, int64_t* __restrict__ seg_hi,
                                                             int32_t* __restrict__ status) {
  using Scan = rocprim::block_scan<int, 1024>;
  using Sort = rocprim::block_radix_sort<unsigned, 1024, 16>;
  __shared__ union {
    unsigned all[kSegSmallMax];  // the element-wise result
    typename Sort::storage_type sort;
    struct {
      unsigned v[kSegRunsMax];            // sorted run heads: key << 12 | run number
      unsigned off[kSegRunsMax + 1];      // output offset of the sorted run
      unsigned short start[kSegRunsMax + 1];  // input position of run number r
      typename Scan::storage_type scan;
    } r;
  } L;
  __shared__ int total_runs;
  const int tid = threadIdx.x;
  auto clamp_key = [&](long long kk, bool& bad) {
    if (kk < 0 || kk >= num_keys) {
      bad = true;
      kk = kk < 0 ? 0 : num_keys - 1;
    }
    return (int)kk;
  };
  bool bad = false;
  // ---- runs: thread t owns the positions [c t, c (t + 1)) ----
  const int c = (n + 1023) >> 10;  // <= 16
  const int lo = tid * c < n ? tid * c : n, hi = lo + c < n ? lo + c : n;
  int kk[16];
  int prev = lo > 0 && lo < n ? clamp_key(key[lo - 1], bad) : -1;
  unsigned heads = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    kk[e] = -1;
    if (lo + e < hi) {
      kk[e] = clamp_key(key[lo + e], bad);
      if (lo + e == 0 || kk[e] != prev) heads |= 1u << e;
      prev = kk[e];
    }
  }
  if (bad) atomicOr(status, TGMX_ST_EDGE_RANGE);
  int first_run;
  Scan().exclusive_scan(__popc(heads), first_run, 0, L.r.scan);
  if (tid == 1023) total_runs = first_run + __popc(heads);
  __syncthreads();
  const int R = total_runs;
  if (R <= kSegRunsMax) {
    int P = 64;
    while (P < R) P <<= 1;
    int rid = first_run;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (heads >> e & 1) {
        L.r.v[rid] = ((unsigned)kk[e] << 12) | (unsigned)rid;
        L.r.start[rid] = (unsigned short)(lo + e);
        ++rid;
      }
    }
    for (int i = R + tid; i < P; i += 1024) L.r.v[i] = 0xffffffffu;
    if (tid == 0) L.r.start[R] = (unsigned short)n;  // n <= 16384 fits
    __syncthreads();
    lds_bitonic(L.r.v, P, tid);
    // output offsets: exclusive scan of the run lengths in sorted order (4 runs per thread)
    int len[4], sum = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = tid * 4 + e;
      len[e] = 0;
      if (q < R) {
        const int id = (int)(L.r.v[q] & 0xfffu);
        len[e] = (int)L.r.start[id + 1] - (int)L.r.start[id];
      }
      sum += len[e];
    }
    int base;
    __syncthreads();  // (the scan's storage is next to, not inside, what was just read -- but the previous scan's use of it is over)
    Scan().exclusive_scan(sum, base, 0, L.r.scan);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = tid * 4 + e;
      if (q < R) L.r.off[q] = (unsigned)base;
      base += len[e];
    }
    if (tid == 0) L.r.off[R] = (unsigned)n;
    __syncthreads();
    // every output position finds its run: the last sorted run whose offset is <= q
    for (int q = tid; q < n; q += 1024) {
      int a = 0, b = R;  // off[a] <= q < off[b]
      while (b - a > 1) {
        const int mid = (a + b) >> 1;
        if (L.r.off[mid] <= (unsigned)q) a = mid;
        else b = mid;
      }
      const int id = (int)(L.r.v[a] & 0xfffu);
      order[q] = (long long)L.r.start[id] + (q - (int)L.r.off[a]);
    }
    // bounds as searchsorted gives them: a key without entries gets lo = hi = the position where it would be inserted
    for (int q = tid; q < R; q += 1024) {
      const int kq = (int)(L.r.v[q] >> 12);
      const int kp = q == 0 ? -1 : (int)(L.r.v[q - 1] >> 12);
      const long long at = L.r.off[q];
      if (kp != kq) {
        seg_lo[kq] = at;
        for (int g = kp + 1; g < kq; ++g) seg_lo[g] = seg_hi[g] = at;  // the absent keys just below this one
      }
      if (q == R - 1 || (int)(L.r.v[q + 1] >> 12) != kq) seg_hi[kq] = L.r.off[q + 1];
    }
    const int last = (int)(L.r.v[R - 1] >> 12);
    for (int g = last + 1 + tid; g < num_keys; g += 1024) seg_lo[g] = seg_hi[g] = n;
    return;
  }
  // ---- keys without run structure: a stable radix sort of the packed values (key << 14 | index) over the KEY bits only, the
  // elements in registers in input order (thread t holds [c t, c (t + 1)); pads sort behind everything)
  __syncthreads();  // the scan's storage lies inside the union
  unsigned x[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) x[e] = lo + e < hi ? ((unsigned)kk[e] << 14) | (unsigned)(lo + e) : 0xffffffffu;
  unsigned bits = 1;
  while ((1 << bits) < num_keys) ++bits;
  Sort().sort(x, L.sort, 14u, 14u + bits > 31u ? 32u : 14u + bits + 1u);  // (+1: the pads' all-ones key sorts last)
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) L.all[tid * 16 + e] = x[e];
  __syncthreads();
  const unsigned* v = L.all;
  for (int i = tid; i < n; i += 1024) {
    const unsigned x = v[i];
    const int kq = (int)(x >> 14);
    order[i] = (long long)(x & 0x3fffu);
    const int kp = i == 0 ? -1 : (int)(v[i - 1] >> 14);
    if (kp != kq) {
      seg_lo[kq] = i;
      for (int g = kp + 1; g < kq; ++g) seg_lo[g] = seg_hi[g] = i;  // the absent keys just below this one
    }
    if (i == n - 1 || (int)(v[i + 1] >> 14) != kq) seg_hi[kq] = i + 1;
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
  // measured (TGN batch, ~13 k edges whose targets are the NEIGHBOURS, so no run structure): the one-workgroup kernel takes ~49 us
  // (block radix sort; 58 with the bitonic network) against 43 us for the rocPRIM chain -- one CU against the chip -- so it is off
  // unless TGMX_SEGSORT_SMALL=1 (read per call: the tests switch it); seed-major targets (run path, ~12 us) would want it on
  const char* knob = getenv("TGMX_SEGSORT_SMALL");
  const bool small_on = knob && atoi(knob) != 0;
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

