// Distinct (node id, time) pairs of one level of the hop tree.
//
// Hop h + 1 of the recency sampler is seeded with hop h's output slots (tgm/hooks/neighbors/recency.py:141-143: the flattened
// nbr_nids / nbr_edge_time, pads included), and a seed's window is a function of (id, time) and of the sampler's state -- which no
// lookup of the call modifies (the update runs after all of them, recency.py:161-163).  So two slots with the same (id, time) are
// the SAME row of every deeper level: same window, same edge ids, same layer outputs (tgm/nn/encoder/tgat.py:128-136 computes
// them per slot all the same).  At the headline shape (12 000 hop-1 seeds) 35-60 % of the slots are pads (one pair: (-1, 0)) and a
// quarter of the rest repeat (hubs, both endpoints of an edge, a user's burst of events).  tgmx_tgat_forward computes each distinct
// row once; this file finds them: an open-addressing table keyed by the pair, the owner of an entry = the smallest slot index that
// carries the pair (deterministic), numbered in wave order.
#include "common.h"

namespace tgmx {

constexpr int kPairEmpty = 0x7f7f7f7f;  // what hipMemsetAsync(0x7f) leaves in an int32

__device__ __forceinline__ unsigned pair_hash(int id, long long t) {
  unsigned long long x = ((unsigned long long)(unsigned)id << 32) ^ (unsigned long long)t ^ ((unsigned long long)t >> 29);
  x *= 0x9E3779B97F4A7C15ull;
  x ^= x >> 32;
  x *= 0xD1B54A32D192ED03ull;
  return (unsigned)(x >> 32);
}

// table[cap] (cap a power of two, every entry kPairEmpty on entry): afterwards entry slot_of[r] holds the smallest row index that
// carries row r's pair.
__global__ __launch_bounds__(256) void pair_insert_kernel(const int32_t* __restrict__ ids, const int64_t* __restrict__ ts, int n, int* table,
                                                          unsigned mask, int* __restrict__ slot_of, int* __restrict__ count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) *count = 0;  // (the numbering kernel behind this one adds to it)
  const bool active = r < n;
  const int id = active ? ids[r] : 0;
  const long long t = active ? ts[r] : 0;
  // pad slots are a third to two thirds of a level and all carry one pair: the wave sends its lowest pad lane, the others copy
  const int lane = lane_id();
  const bool pad = active && id < 0;
  const unsigned long long pads = __ballot(pad);
  int leader = -1;
  bool follow = false;
  if (pads) {
    leader = __ffsll((long long)pads) - 1;
    const int lid = __shfl(id, leader);
    const long long lt = __shfl(t, leader);
    follow = pad && lane != leader && id == lid && t == lt;
  }
  int slot = -1;
  if (active && !follow) {
    unsigned s = pair_hash(id, t) & mask;
    for (;;) {
      int cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == kPairEmpty) {
        const int prev = atomicCAS(&table[s], kPairEmpty, r);
        if (prev == kPairEmpty) break;
        cur = prev;
      }
      // an entry's owner changes only among rows of ONE pair, so whichever owner was read decides whose entry this is
      if (ids[cur] == id && ts[cur] == t) {
        atomicMin(&table[s], r);
        break;
      }
      s = (s + 1) & mask;
    }
    slot = (int)s;
  }
  if (pads) {
    const int ls = __shfl(slot, leader);
    if (follow) slot = ls;
  }
  if (active) slot_of[r] = slot;
}

// owner[r] = the representative of row r's pair; representatives are numbered 0 .. count - 1 (cidx[rep] = its number,
// uniq[number] = rep).  Numbers are handed out per wave (one atomic per wave): dense, the order across waves is unspecified.
__global__ __launch_bounds__(256) void pair_number_kernel(int n, const int* __restrict__ table, const int* __restrict__ slot_of, int* __restrict__ owner,
                                                          int* __restrict__ cidx, int* __restrict__ uniq, int* count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = r < n;
  const int own = active ? table[slot_of[r]] : -1;
  if (active) owner[r] = own;
  const bool rep = active && own == r;
  const unsigned long long reps = __ballot(rep);
  if (!reps) return;
  const int lane = lane_id();
  int base = 0;
  if (lane == 0) base = atomicAdd(count, __popcll(reps));
  base = __shfl(base, 0);
  if (rep) {
    const int c = base + __popcll(reps & ((1ull << lane) - 1ull));
    cidx[r] = c;
    uniq[c] = r;
  }
}

static inline unsigned pair_table_cap(long long n) {
  unsigned cap = 1024;
  while ((long long)cap < 2 * n) cap <<= 1;
  return cap;
}

}  // namespace tgmx

using namespace tgmx;

extern "C" size_t tgmx_pair_dedup_workspace_bytes(int64_t n) {
  if (n < 0 || n > (1ll << 28)) return 0;
  return ((size_t)pair_table_cap(n) + (size_t)(n > 0 ? n : 1)) * sizeof(int32_t);
}

extern "C" int tgmx_pair_dedup(const int32_t* ids, const int64_t* times, int64_t n, int32_t* uniq, int32_t* owner, int32_t* cidx, int32_t* count,
                               void* workspace, size_t workspace_bytes, tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0 && n <= (1ll << 28), "pair_dedup: n=%lld outside [0, 2^28]", (long long)n);
  TGMX_REQUIRE(count, "pair_dedup: null count");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (hipMemsetAsync(count, 0, sizeof(int32_t), st) != hipSuccess) {
      set_error("pair_dedup: memset failed");
      return TGMX_E_LAUNCH;
    }
    return TGMX_OK;
  }
  TGMX_REQUIRE(ids && times && uniq && owner && cidx && workspace, "pair_dedup: null pointer");
  TGMX_REQUIRE(workspace_bytes >= tgmx_pair_dedup_workspace_bytes(n), "pair_dedup: workspace too small (%zu < %zu)", workspace_bytes,
               tgmx_pair_dedup_workspace_bytes(n));
  TGMX_REQUIRE(((uintptr_t)workspace & 3) == 0, "pair_dedup: workspace must be 4-byte aligned");
  const unsigned cap = pair_table_cap(n);
  int* table = reinterpret_cast<int*>(workspace);
  int* slot_of = table + cap;
  if (hipMemsetAsync(table, 0x7f, (size_t)cap * sizeof(int), st) != hipSuccess) {
    set_error("pair_dedup: memset failed");
    return TGMX_E_LAUNCH;
  }
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(pair_insert_kernel, dim3(blocks), dim3(256), 0, st, ids, times, (int)n, table, cap - 1, slot_of, count);
  TGMX_CHECK_LAUNCH("pair_insert");
  hipLaunchKernelGGL(pair_number_kernel, dim3(blocks), dim3(256), 0, st, (int)n, table, slot_of, owner, cidx, uniq, count);
  TGMX_CHECK_LAUNCH("pair_number");
  return TGMX_OK;
}
