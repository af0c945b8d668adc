// DGData.discretize's grouping (tgm/data/dg_data.py:471-500, `_get_keep_indices`) on the device.
//
// Per event group (edges / node events / node labels): bucket = floor(float64(t) * factor) as int32, a radix key
// [bucket, src, dst] evaluated in the reference's int32 arithmetic (id_key = src * base + dst with base = ids.max() + 1,
// final_key = bucket * (id_key.max() + 1) + id_key -- int32 tensors: the products wrap exactly like torch's), a STABLE
// sort of the keys (rocPRIM LSD radix sort), first-of-group marks, and the kept event positions in ascending
// (chronological) order.  The reference's second sort (`keep.sort()`) is replaced by scattering the first-of-group
// marks back to event order and compacting them (rocprim::select over a counting iterator): O(n), already ordered.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"

namespace tgmx {

struct DiscArgs {
  const int64_t* time;   // [n] event times of this group
  const int32_t* id0;    // [n] src (edges) or node id
  const int32_t* id1;    // [n] dst (edges) or NULL
  double factor;
  long long n;
  int32_t* bucket;       // [n] out
  int32_t* red;          // [2]: max(ids) (both columns), max(id_key)
  int32_t* id_key;       // [n] scratch
  unsigned int* key_in;  // [n] sign-flipped final keys
  unsigned int* val_in;  // [n] event index
  unsigned int* key_out;
  unsigned int* val_out;
  unsigned char* keep;   // [n] first-of-group mark per EVENT (event order)
};

__global__ __launch_bounds__(256) void disc_bucket_max_kernel(const DiscArgs a) {
  int mx = -2147483647 - 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
    a.bucket[i] = (int)floor((double)a.time[i] * a.factor);  // .to(float64) * factor, .floor().int()
    const int v0 = a.id0[i];
    mx = v0 > mx ? v0 : mx;
    if (a.id1) {
      const int v1 = a.id1[i];
      mx = v1 > mx ? v1 : mx;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(mx, o);
    mx = other > mx ? other : mx;
  }
  if (lane_id() == 0) atomicMax(&a.red[0], mx);
}

__global__ __launch_bounds__(256) void disc_idkey_kernel(const DiscArgs a) {
  const unsigned base = (unsigned)(a.red[0] + 1);
  int mx = -2147483647 - 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
    // int32 tensor arithmetic wraps (two's complement): evaluate in unsigned, reinterpret
    const int key = a.id1 ? (int)((unsigned)a.id0[i] * base + (unsigned)a.id1[i]) : a.id0[i];
    a.id_key[i] = key;
    mx = key > mx ? key : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(mx, o);
    mx = other > mx ? other : mx;
  }
  if (lane_id() == 0) atomicMax(&a.red[1], mx);
}

__global__ __launch_bounds__(256) void disc_final_key_kernel(const DiscArgs a) {
  const unsigned base = (unsigned)(a.red[1] + 1);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
    const int key = (int)((unsigned)a.bucket[i] * base + (unsigned)a.id_key[i]);
    a.key_in[i] = (unsigned)key ^ 0x80000000u;  // signed order as unsigned order
    a.val_in[i] = (unsigned)i;
  }
}

__global__ __launch_bounds__(256) void disc_mark_kernel(const DiscArgs a) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < a.n; p += (long long)gridDim.x * blockDim.x)
    a.keep[a.val_out[p]] = (p == 0 || a.key_out[p] != a.key_out[p - 1]) ? 1 : 0;
}

struct DiscLayout {
  size_t red, id_key, key_in, val_in, key_out, val_out, keep, temp, temp_bytes, total;
};
static int disc_layout(long long n, DiscLayout& w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  };
  const size_t m = (size_t)(n > 0 ? n : 1);
  w.red = take(256);
  w.id_key = take(m * 4);
  w.key_in = take(m * 4);
  w.val_in = take(m * 4);
  w.key_out = take(m * 4);
  w.val_out = take(m * 4);
  w.keep = take(m);
  size_t sort_bytes = 0, sel_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, m, 0, 32,
                                (hipStream_t) nullptr) != hipSuccess)
    return TGMX_E_LAUNCH;
  if (rocprim::select(nullptr, sel_bytes, rocprim::counting_iterator<long long>(0), (unsigned char*)nullptr, (long long*)nullptr,
                      (long long*)nullptr, m, (hipStream_t) nullptr) != hipSuccess)
    return TGMX_E_LAUNCH;
  w.temp_bytes = sort_bytes > sel_bytes ? sort_bytes : sel_bytes;
  w.temp = take(w.temp_bytes);
  w.total = off + 256;
  return TGMX_OK;
}

}  // namespace tgmx

using namespace tgmx;

extern "C" size_t tgmx_discretize_workspace_bytes(int64_t n) {
  DiscLayout w;
  if (n < 0 || n >= (1ll << 31) || disc_layout(n, w) != TGMX_OK) return 0;
  return w.total;
}

extern "C" int tgmx_discretize_keep(const int64_t* time, const int32_t* id0, const int32_t* id1, int64_t n, double factor, int32_t* bucket,
                                    int64_t* keep_pos, int64_t* keep_count, void* workspace, size_t workspace_bytes, tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0 && n < (1ll << 31), "discretize_keep: n=%lld", (long long)n);
  TGMX_REQUIRE(keep_count, "discretize_keep: null count");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    (void)hipMemsetAsync(keep_count, 0, sizeof(int64_t), st);
    return TGMX_OK;
  }
  TGMX_REQUIRE(time && id0 && bucket && keep_pos && workspace, "discretize_keep: null pointer");
  DiscLayout w;
  if (disc_layout(n, w) != TGMX_OK) {
    set_error("discretize_keep: rocPRIM size query failed");
    return TGMX_E_LAUNCH;
  }
  TGMX_REQUIRE(workspace_bytes >= w.total, "discretize_keep: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  char* base = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  DiscArgs a{};
  a.time = time; a.id0 = id0; a.id1 = id1; a.factor = factor; a.n = n; a.bucket = bucket;
  a.red = reinterpret_cast<int32_t*>(base + w.red);
  a.id_key = reinterpret_cast<int32_t*>(base + w.id_key);
  a.key_in = reinterpret_cast<unsigned*>(base + w.key_in);
  a.val_in = reinterpret_cast<unsigned*>(base + w.val_in);
  a.key_out = reinterpret_cast<unsigned*>(base + w.key_out);
  a.val_out = reinterpret_cast<unsigned*>(base + w.val_out);
  a.keep = reinterpret_cast<unsigned char*>(base + w.keep);
  (void)hipMemsetAsync(a.red, 0x80, 2 * sizeof(int32_t), st);  // 0x80808080: below every id this path accepts (ids >= 0)
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const dim3 grid((unsigned)blocks), block(256);
  hipLaunchKernelGGL(disc_bucket_max_kernel, grid, block, 0, st, a);
  hipLaunchKernelGGL(disc_idkey_kernel, grid, block, 0, st, a);
  hipLaunchKernelGGL(disc_final_key_kernel, grid, block, 0, st, a);
  size_t temp_bytes = w.temp_bytes;
  if (rocprim::radix_sort_pairs(base + w.temp, temp_bytes, a.key_in, a.key_out, a.val_in, a.val_out, (size_t)n, 0, 32, st) != hipSuccess) {
    set_error("discretize_keep: radix sort failed");
    return TGMX_E_LAUNCH;
  }
  hipLaunchKernelGGL(disc_mark_kernel, grid, block, 0, st, a);
  temp_bytes = w.temp_bytes;
  if (rocprim::select(base + w.temp, temp_bytes, rocprim::counting_iterator<long long>(0), a.keep, reinterpret_cast<long long*>(keep_pos),
                      reinterpret_cast<long long*>(keep_count), (size_t)n, st) != hipSuccess) {
    set_error("discretize_keep: select failed");
    return TGMX_E_LAUNCH;
  }
  TGMX_CHECK_LAUNCH("discretize_keep");
  return TGMX_OK;
}
