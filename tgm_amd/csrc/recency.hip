// k-most-recent temporal neighbor lookup (CSR index and streaming rings) for gfx950.
//
// Work decomposition: ONE 64-lane wave per seed.
//   phase A (index, latency bound, tiny traffic): the wave locates the seed's
//     observable window of <= B adjacency records with a 64-ary cooperative
//     search (CSR) or reads its ring row (streaming), loads the window as one
//     coalesced dwordx4-per-lane access, finds the rightmost entry with
//     ts < q by ballot, and redistributes the k winners with lane shuffles.
//   phase B (gather, HBM bound, ~all of the bytes): the seed's [k, D] output
//     block is contiguous, so the wave streams it as flat 16-byte vectors --
//     every store instruction writes 1 KiB contiguous, every load reads the
//     matching 16 bytes of edge_x[eid[slot]] (row pieces are contiguous).
//     The [S, B, D] intermediate the reference materialises
//     (tgm/hooks/neighbors/recency.py:258) never exists.
// Rows are copied, never recomputed, so features are bit-exact by construction.
//
// File map (one translation unit: the lookup launches carry pieces of the ring update, so both live here):
//   UpdateArgs, keys, the single-workgroup sort / placement body (update_block_body)
//   riders: update_chunk_sort, update_merge_riding, rider_barrier, update_side_work -- the state-independent half of
//     the ring update as workgroup-sized pieces that ride inside the lookup launches
//   lookup pieces (fetch_seed ... lookup_seed) and the kernels built from them: recency_lookup_kernel (one hop),
//     recency_lookup_fused01_kernel (hop 0 + hop 1 in one launch), lookup_packed_kernel (narrow rows)
//   the stand-alone update paths (one workgroup / chunk sort + merge / rocPRIM radix sort), uniform sampler, C entry points
#include <atomic>
#include <cstdlib>
#include <cstring>

#include <hip/hip_ext.h>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/block/block_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"

namespace tgmx {

struct __align__(16) Rec {
  int nbr;
  int eid;
  long long ts;
};
static_assert(sizeof(Rec) == sizeof(tgmx_adj_t), "record layout");

// hop-0 seeds given as groups (recency.py:173-237 concatenates them with torch.cat): the lookup wave of seed s
// reads its (node, time) from the group that covers s and publishes the concatenation as a side effect
struct SeedGroups {
  const int32_t* nid[TGMX_MAX_SEED_GROUPS];
  const int64_t* ts[TGMX_MAX_SEED_GROUPS];
  long long end[TGMX_MAX_SEED_GROUPS];  // exclusive end offset of group g in the concatenation
  int32_t* out_nid;
  int64_t* out_ts;
  int groups;  // 0: seeds / qtimes are read instead
  // group `gen` (-1: none) has no id array: its ids are RandomNegativeEdgeSamplerHook's draws (negative_draw), generated
  // where they are needed -- every wave that fetches seed s re-derives the same id -- and published to gen_out /
  // gen_out_ts (the hook's `neg` / `neg_time`) by the hop-0 wave of that seed
  int gen;
  int gen_low;
  unsigned gen_range;
  unsigned long long gen_seed, gen_call, gen_index0;
  int32_t* gen_out;
  int64_t* gen_out_ts;
};

struct LookupArgs {
  SeedGroups grp;
  const int64_t* indptr;    // CSR
  const Rec* recs;          // CSR adjacency or ring rows
  const int32_t* write_pos; // ring
  const float* edge_x;
  const int32_t* seeds;
  const int64_t* qtimes;
  int32_t* out_nid;
  int64_t* out_ts;
  float* out_x;
  int32_t* status;
  long long S;
  long long ev_lo, ev_hi;
  int D, k, B, N, allow_pad;
  int row_vecs;  // D / VEC
  FastDiv dv;    // division by row_vecs
  FastDiv dv_seed;  // division by k * row_vecs (tile kernel: flat piece index -> seed of the tile)
  int32_t* out_eid;   // optional [S, k]: the edge id behind every output slot (-1: pad) -- with out_x == NULL the features are not
  int32_t* out_eid1;  //   copied at all and the consumer gathers them from the resident store by id (fused launch: hop 1's)
  int x_by_pos;       // static index: feature rows are stored in ADJACENCY order (row = record position), not addressed by eid
  long long* cursor;  // static index, optional: per node, where its visible prefix ended the last time it was looked up (a hint)
  int leave_room;   // tile kernel: other launches must fit beside this one (the large ring update's chain on the side stream)
  // ring update work riding along (tgmx_recency_step): the first side_blocks workgroups run side_stage on the
  // launch's second argument instead of looking anything up
  unsigned side_blocks;
  int side_stage;
  // tail commit (tgmx_recency_step, placement decided by the rider): the LAST tail_blocks workgroups wait until every
  // lookup workgroup and the rider have finished, then write the batch into the rings -- the update's write-back as the
  // tail of the lookup launch instead of a launch of its own
  unsigned tail_blocks;
  // fused hop 0 + hop 1 launch: hop 1's k and outputs (rows of hop 1 = S * k)
  int k1;
  int32_t* out_nid1;
  int64_t* out_ts1;
  float* out_x1;
  // DELTA feature writes (persistent output buffers, tgmx_recency_step_t.out_valid): out_valid[s] = the SPAN row s of out_x still
  // holds from the previous call -- the slots from its leftmost non-pad one to the end; everything left of that is zero.  The row
  // then needs writes only from slot k - max(old span, new span) on: the pads left of that are zeros already.  NULL: every slot.
  int32_t* out_valid;
  int32_t* out_valid1;
  int32_t* out_valid_prev;   // optional: receives the span a row held BEFORE this call (byte accounting of a timed launch)
  int32_t* out_valid_prev1;
};

template <int VEC>
struct VecOf;
template <>
struct VecOf<4> { using type = float4; };
template <>
struct VecOf<2> { using type = float2; };
template <>
struct VecOf<1> { using type = float; };

template <typename T>
__device__ __forceinline__ T zero_vec();
template <>
__device__ __forceinline__ float4 zero_vec<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <>
__device__ __forceinline__ float2 zero_vec<float2>() { return make_float2(0.f, 0.f); }
template <>
__device__ __forceinline__ float zero_vec<float>() { return 0.f; }

// Number of records in [a, z) whose eid < bound, given that those records form
// a prefix (true for batch-boundary bounds).  64 probes/round.
__device__ __forceinline__ long long wave_prefix_count(const Rec* recs, long long a, long long z,
                                                       long long bound, int lane) {
  long long lo = a, hi = z;
  while (hi - lo > kWave) {
    const long long len = hi - lo;
    const long long stride = (len + kWave - 1) / kWave;
    long long idx = lo + (long long)(lane + 1) * stride - 1;
    if (idx > hi - 1) idx = hi - 1;
    const bool less = (long long)recs[idx].eid < bound;
    const int c = __popcll(__ballot(less));
    long long nlo = lo, nhi = hi;
    if (c > 0) {
      long long p = lo + (long long)c * stride - 1;
      if (p > hi - 1) p = hi - 1;
      nlo = p + 1;
    }
    if (c < kWave) {
      long long p = lo + (long long)(c + 1) * stride - 1;
      if (p > hi - 1) p = hi - 1;
      nhi = p;  // recs[p] is >= bound: the answer is <= p
    }
    lo = nlo;
    hi = nhi < nlo ? nlo : nhi;
  }
  const long long idx = lo + lane;
  const bool less = idx < hi && (long long)recs[idx].eid < bound;
  return lo + __popcll(__ballot(less)) - a;
}

// ---------------------------------------------------------------------------
// Streaming rings: batched append -- a faithful restatement of
// tgm/hooks/neighbors/recency.py:323-399 INCLUDING its key arithmetic.
//
// Entry j < n is (src[j] -> dst[j]), entry n + j is (dst[j] -> src[j]).  The
// reference stably argsorts key = node * (max_t + 1) + t, where the product is
// evaluated in int32 (an int32 tensor times a 0-dim int64 tensor stays int32) and
// therefore wraps at dataset scale; it then treats every RUN of equal node ids
// in that order as one group: keeps the run's last B entries, scatters them at
// (write_pos[node] + rank_in_run) % B (runs of one node collide: the later one in
// sorted order wins) and advances write_pos by the number of kept entries.
// key_wrap32 = 0 evaluates the key in int64 (the intended (node, time) order).
//
// The sort of the m = 2 * batch_size entries is an all-pairs rank: exact, stable,
// deterministic, no atomics; O(m^2 / lanes), microseconds for m <= ~10^4.
//   k1 sort : rank every entry, scatter (entry, node) to its sorted position
//   k2 place: run boundaries -> keep / ring slot per sorted position
//   k3 write: resolve slot collisions (last wins), write records, commit write_pos
//   k4 feat : one wave per written record copies its D-float feature row
// ---------------------------------------------------------------------------
struct UpdateArgs {
  Rec* ring;
  int32_t* write_pos;
  float* ring_x;        // [N*B, D] feature row of every ring slot
  const float* edge_x;  // [n, D] rows of this batch (null -> zeros)
  const int32_t* src;
  const int32_t* dst;
  const int64_t* ts;
  int32_t* sorted_j;     // scratch [m]: entry index at sorted position p
  int32_t* sorted_node;  // scratch [m]: its node (-1 = invalid entry)
  int32_t* target;       // scratch [m]: ring row it is placed at (-1 = dropped)
  int32_t* winner;       // scratch [m]: ring row it finally owns (-1 = none)
  Rec* sorted_rec;       // scratch [m] (chunked path): the record of the entry at sorted position p
  // chunked path (1024 < m <= 4096):
  long long* key;        // scratch: chunk-sorted keys
  int32_t* node;         // scratch: chunk-sorted entry indices
  // large-batch path (m > 4096):
  long long* span;             // scratch: max(ts) + 1
  unsigned long long* keys_in; // scratch [m]: radix keys of the entries
  unsigned int* vals_in;       // scratch [m]: entry indices
  int32_t* hash_key;           // scratch [2^hash_bits]: ring rows placed on (-1 empty)
  int32_t* hash_maxp;          // scratch [2^hash_bits]: last sorted position placed there
  int32_t* run_start;          // scratch [m]: first sorted position of p's run
  int32_t* run_len;            // scratch [m]: run length, at the run's first position
  int hash_bits;
  long long ts_bound;   // > 0: every timestamp is promised to lie in [0, ts_bound] (bounded-bit radix sort)
  int sorted_ts;        // != 0: the batch's timestamps are promised non-decreasing (large path: span = ts[n - 1] + 1)
  int sort_bits;        // large path: number of low key bits the radix sort looks at (64 = all)
  int32_t* barrier;  // 2 ints at the head of the caller's scratch: the riders' self-resetting barrier (count, generation)
  int32_t* status;
  int guard_mask;    // != 0: kernels that write ring state return untouched when *status has one of these bits (bad seeds seen by
                     // the lookups of the same call): the reference validates before it changes anything (recency.py:173-237)
  long long n, m, eid0;
  int B, N, D, key_wrap32;
};

__device__ __forceinline__ long long update_key(int node, long long t, long long span, int wrap32) {
  if (wrap32) {
    const unsigned int prod = (unsigned int)node * (unsigned int)(int)span;  // int32 multiply, two's complement wrap
    return (long long)(int)prod + t;
  }
  return (long long)node * span + t;
}

__device__ __forceinline__ void update_entry(const UpdateArgs& a, long long j, int& node, int& nbr, long long& t,
                                             long long& i) {
  const bool rev = j >= a.n;
  i = rev ? j - a.n : j;
  const int s = a.src[i], d = a.dst[i];
  node = rev ? d : s;
  nbr = rev ? s : d;
  t = a.ts[i];
}

constexpr int kBlockMaxM = 4096;  // largest batch (entries) of the single-workgroup placement kernel
constexpr int kChunk = 256;       // entries per workgroup of the spread sort

// max over ts[0, n) of this thread's strided share, 8 independent loads in flight per round
__device__ __forceinline__ long long strided_max_ts(const int64_t* __restrict__ ts, long long n, int tid, int nthr) {
  long long mx = -0x7fffffffffffffffLL;
  for (long long base = tid; base < n; base += 8ll * nthr) {
    long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long x = base + (long long)u * nthr;
      v[u] = x < n ? ts[x] : mx;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) mx = v[u] > mx ? v[u] : mx;
  }
  return mx;
}

__device__ __forceinline__ long long strided_min_ts(const int64_t* __restrict__ ts, long long n, int tid, int nthr) {
  long long mn = 0x7fffffffffffffffLL;
  for (long long base = tid; base < n; base += 8ll * nthr) {
    long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long x = base + (long long)u * nthr;
      v[u] = x < n ? ts[x] : mn;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) mn = v[u] < mn ? v[u] : mn;
  }
  return mn;
}

// Batches of up to kBlockMaxM entries (every TGB-style batch size, and the replicated update of an 8-rank
// global batch): ONE workgroup with the working set in LDS (gfx950: 160 KiB per CU).
//   sort  : bitonic network over (key, entry index) pairs -- unique, so the order is the stable one.  Thread t
//           holds E consecutive elements in registers: compare distances < E are in-register, distances < 64 E
//           are lane shuffles inside the wave (no barrier, no LDS), only distances >= 64 E go through LDS
//           (6 of the 45 steps at m = 400; 10 of 78 at m = 4096).
//   runs  : maximal runs of equal node in sorted order by an inclusive max-scan (hub runs are long).
//   place : ring slot = (write_pos[node] % B + offset inside the kept part of the run) % B; collisions between
//           runs of one node are resolved through an LDS open-addressing hash (atomicMax of the sorted position:
//           the last one wins).
//   write : winners write their record; the last entry of every run advances write_pos by #kept with one
//           atomicAdd (every reader takes write_pos % B).
constexpr int kBlockThreads = 1024;


// element e (mine) against its partner at distance jj in the step of bitonic stage k: keep min or max
__device__ __forceinline__ void bitonic_select(long long& key, int& pay, long long pk, int pp, int e, int jj, int k) {
  const bool low = (e & jj) == 0, asc = (e & k) == 0;
  const bool mine_after = pair_after(key, pay, pk, pp);
  if ((low == asc) == mine_after) {  // keep-min and mine is larger, or keep-max and mine is smaller
    key = pk;
    pay = pp;
  }
}

__device__ __forceinline__ bool update_blocked(const UpdateArgs& a) {
  return a.guard_mask != 0 && (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & a.guard_mask) != 0;
}

__device__ __forceinline__ void commit_write_pos(int32_t* wp, int kept, int B) {
  const int old = atomicAdd(wp, kept);
  constexpr int kFold = 1 << 30;
  if (old < kFold && old + kept >= kFold) atomicSub(wp, kFold / B * B);  // stay far from int32 overflow
}

// LDS of the placement (and, stand-alone, of the sort); the riders of the lookup launches carve theirs from a union
template <int MAXM>
struct PlaceLds {
  int sn[MAXM];   // node at sorted position p (-1 invalid)
  int len[MAXM];  // run length, stored at the run's first position
  int h_key[2 * MAXM], h_maxp[2 * MAXM];  // hash load factor <= 0.5
  long long red[kBlockThreads / kWave], red2[kBlockThreads / kWave];
  int wave_tot[kBlockThreads / kWave];
};
template <int MAXM>
struct SortLds {
  long long key[MAXM];
  int pay[MAXM];
};

// device-coherent accesses for what rider workgroups of ONE launch hand to each other (other CUs, other XCDs' L2s):
// sc1 stores / loads reach the device's coherence point, so no cache write-back or invalidate is needed around the
// riders' barrier -- the lookups around them keep their L2 contents
template <typename T>
__device__ __forceinline__ void st_agent(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T ld_agent(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// DEFER: decide everything, write nothing to the rings -- winner[p] = the ring row position p finally owns (-1 none),
// target[p] = the write_pos increment position p commits (0 none); `ring_update_commit_kernel` applies them.
// COH: the presorted arrays were written by OTHER workgroups of this launch (riders): read them device-coherently
template <int E, int MAXM, bool PRESORTED, bool DEFER, bool COH = false>
__device__ __forceinline__ void update_block_body(const UpdateArgs& a, PlaceLds<MAXM>& L, SortLds<PRESORTED ? 1 : MAXM>& S,
                                                  int part = 0, int parts = 1) {
  constexpr int H = 2 * MAXM;
  constexpr int HBITS = MAXM == 512 ? 10 : (MAXM == 1024 ? 11 : (MAXM == 2048 ? 12 : 13));
  static_assert((1 << HBITS) == H, "hash size");
  long long* s_key = S.key;
  int* s_pay = S.pay;
  int* s_sn = L.sn;
  int* s_len = L.len;
  int* h_key = L.h_key;
  int* h_maxp = L.h_maxp;
  long long* red = L.red;
  long long* red2 = L.red2;
  int* wave_tot = L.wave_tot;
  const int m = (int)a.m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwaves = nthr >> 6;
  const int P = nthr * E;  // power of two >= m; thread t owns sorted positions t*E .. t*E + E-1
  const bool blocked = DEFER ? false : update_blocked(a);  // immediate writers only; deferred decisions are applied by a guarded commit

  for (int x = tid; x < H; x += nthr) {
    h_key[x] = -1;
    h_maxp[x] = -1;
  }
  int pay[E], node[E], w[E];
  if constexpr (PRESORTED) {
    // sorted order, nodes and write_pos % B were produced by the chunk-sort + merge kernels
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const int p = tid * E + r;
      pay[r] = p;
      node[r] = -1;
      w[r] = 0;
      if (p < m) {
        pay[r] = COH ? ld_agent(&a.sorted_j[p]) : a.sorted_j[p];
        node[r] = COH ? ld_agent(&a.sorted_node[p]) : a.sorted_node[p];
        w[r] = COH ? ld_agent(&a.target[p]) : a.target[p];
      }
    }
  } else {
    static_assert(PRESORTED || E == 1, "the in-kernel sort holds one element per thread");
    long long mx = strided_max_ts(a.ts, a.n, tid, nthr), mn = strided_min_ts(a.ts, a.n, tid, nthr);
    for (int off = 32; off > 0; off >>= 1) {
      const long long o = __shfl_xor(mx, off), o2 = __shfl_xor(mn, off);
      mx = o > mx ? o : mx;
      mn = o2 < mn ? o2 : mn;
    }
    if (lane == 0) {
      red[wave] = mx;
      red2[wave] = mn;
    }
    __syncthreads();
    mx = red[0];
    mn = red2[0];
    for (int wv = 1; wv < nwaves; ++wv) {
      mx = red[wv] > mx ? red[wv] : mx;
      mn = red2[wv] < mn ? red2[wv] : mn;
    }
    const long long span = mx + 1;
    const bool packed = can_pack(a.key_wrap32, mn, mx);

    long long key = 0x7fffffffffffffffLL;  // padding sorts to the end
    pay[0] = tid;
    if (tid < m) {
      int nd, nbr;
      long long t, i;
      update_entry(a, tid, nd, nbr, t, i);
      key = update_key(nd, t, span, a.key_wrap32);
      if (packed) key = packed_key(key, tid);
    }
    if (packed) bitonic_sort_one<true>(key, pay[0], s_key, s_pay, tid, P);
    else bitonic_sort_one<false>(key, pay[0], s_key, s_pay, tid, P);
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const int p = tid * E + r;
      node[r] = -1;
      w[r] = 0;
      if (p < m) {
        int nd, nbr;
        long long t, i;
        update_entry(a, pay[r], nd, nbr, t, i);
        const bool valid = nd >= 0 && nd < a.N && nbr >= 0 && nbr < a.N;
        if (!valid) atomicOr(a.status, TGMX_ST_EDGE_RANGE);
        if (valid) {
          node[r] = nd;
          w[r] = a.write_pos[nd] % a.B;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < E; ++r) s_sn[tid * E + r] = node[r];
  __syncthreads();

  // run start of every sorted position = last position <= p that opens a run: max-scan, thread -> wave -> block
  int st[E];
  {
    int run = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const int p = tid * E + r;
      if (p == 0 || s_sn[p - 1] != node[r]) run = p;
      st[r] = run;
    }
    int incl = run;  // positions only grow, so the thread's last value is its maximum
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl = o > incl ? o : incl;
    }
    if (lane == kWave - 1) wave_tot[wave] = incl;
    int before = __shfl_up(incl, 1);
    if (lane == 0) before = 0;
    __syncthreads();
    for (int wv = 0; wv < wave; ++wv) before = wave_tot[wv] > before ? wave_tot[wv] : before;
#pragma unroll
    for (int r = 0; r < E; ++r) st[r] = st[r] > before ? st[r] : before;
  }
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int p = tid * E + r;
    if (p < m && (p == m - 1 || s_sn[p + 1] != node[r])) s_len[st[r]] = p - st[r] + 1;
  }
  __syncthreads();

  // placement; collisions between runs of one node resolved by atomicMax of the sorted position
  int tgt[E], hs[E], cnt[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int p = tid * E + r;
    tgt[r] = -1;
    hs[r] = 0;
    cnt[r] = 0;
    if (p < m && node[r] >= 0 && (parts == 1 || node[r] % parts == part)) {
      cnt[r] = s_len[st[r]];
      const int pos = p - st[r];
      const int drop = cnt[r] > a.B ? cnt[r] - a.B : 0;
      if (pos >= drop) {
        const int t = node[r] * a.B + (w[r] + pos - drop) % a.B;
        unsigned h = ((unsigned)t * 2654435761u) >> (32 - HBITS);
        for (;;) {
          const int old = atomicCAS(&h_key[h], -1, t);
          if (old == -1 || old == t) break;
          h = (h + 1) & (H - 1);
        }
        atomicMax(&h_maxp[h], p);
        tgt[r] = t;
        hs[r] = (int)h;
      }
    }
  }
  __syncthreads();

#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int p = tid * E + r;
    if (p >= m) continue;
    if (parts > 1 && (node[r] >= 0 ? node[r] % parts : 0) != part) continue;  // another workgroup's node
    int win = -1, kept = 0;
    if (tgt[r] >= 0) {
      if (h_maxp[hs[r]] == p) {
        Rec rec;
        if constexpr (DEFER) {
          // the commit kernel reads sorted_rec[p] itself
        } else if constexpr (PRESORTED) {
          rec = a.sorted_rec[p];
        } else {
          int nd, nbr;
          long long t, i;
          update_entry(a, pay[r], nd, nbr, t, i);
          rec.nbr = nbr;
          rec.eid = a.eid0 >= 0 ? (int)(a.eid0 + i) : -1;
          rec.ts = t;
        }
        if constexpr (!DEFER) {
          if (!blocked) a.ring[tgt[r]] = rec;
        }
        win = blocked ? -1 : tgt[r];
      }
      if (p == st[r] + cnt[r] - 1) {  // the run's last entry commits the run (every write_pos read is behind a barrier)
        kept = cnt[r] > a.B ? a.B : cnt[r];
        if constexpr (!DEFER) {
          if (!blocked) commit_write_pos(&a.write_pos[node[r]], kept, a.B);
        }
      }
    }
    a.winner[p] = win;
    if constexpr (DEFER) a.target[p] = kept;
    if constexpr (!PRESORTED) a.sorted_j[p] = pay[r];
  }
}

template <int E, int MAXM, bool PRESORTED>
__global__ __launch_bounds__(kBlockThreads) void ring_update_block_kernel(const UpdateArgs a) {
  __shared__ PlaceLds<MAXM> L;
  __shared__ SortLds<PRESORTED ? 1 : MAXM> S;
  // PRESORTED launches may spread the placement over gridDim.x workgroups by node id (every collision is between
  // entries of ONE node): each repeats the run analysis and places / writes only its own nodes
  update_block_body<E, MAXM, PRESORTED, false>(a, L, S, (int)blockIdx.x, (int)gridDim.x);
}

// placement DECISIONS only (PRESORTED, DEFER): reads write_pos, writes winner / target to the scratch -- may run next to the
// lookups of the same batch; ring_update_feat_kernel<COMMIT> applies them afterwards
template <int E, int MAXM>
__global__ __launch_bounds__(kBlockThreads) void ring_update_decide_kernel(const UpdateArgs a) {
  __shared__ PlaceLds<MAXM> L;
  __shared__ SortLds<1> S;
  update_block_body<E, MAXM, true, true>(a, L, S, (int)blockIdx.x, (int)gridDim.x);
}

// ---- the state-independent half of the ring update, as workgroup-sized pieces -------------------------------------
// The order of a batch's entries depends on the batch alone, not on the rings, so `tgmx_recency_step` lets it ride
// along with the lookups: the first `side_blocks` workgroups of the hop-0 launch chunk-sort the entries, the first
// `side_blocks` workgroups of the hop-1 launch merge the chunks (and pre-gather node / write_pos % B / record of every
// sorted position: the rings do not move while lookups run).  What is left after the lookups is the placement kernel
// (PRESORTED) and the feature copy.  The same pieces are the chunked path of the stand-alone `tgmx_ring_update`.

// one 256-thread workgroup sorts entries [chunk * 256, chunk * 256 + 256) by (key, entry index)
struct ChunkSortLds {
  long long key[kChunk];
  int pay[kChunk];
  long long red[kChunk / kWave], red2[kChunk / kWave];
};

template <bool AGENT = false>
__device__ __forceinline__ void update_chunk_sort(const UpdateArgs& a, int chunk, ChunkSortLds& W) {
  long long* s_key = W.key;
  int* s_pay = W.pay;
  long long* red = W.red;
  long long* red2 = W.red2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long mx = strided_max_ts(a.ts, a.n, tid, kChunk), mn = strided_min_ts(a.ts, a.n, tid, kChunk);
  for (int off = 32; off > 0; off >>= 1) {
    const long long o = __shfl_xor(mx, off), o2 = __shfl_xor(mn, off);
    mx = o > mx ? o : mx;
    mn = o2 < mn ? o2 : mn;
  }
  if (lane == 0) {
    red[wave] = mx;
    red2[wave] = mn;
  }
  __syncthreads();
  mx = red[0];
  mn = red2[0];
  for (int w = 1; w < kChunk / kWave; ++w) {
    mx = red[w] > mx ? red[w] : mx;
    mn = red2[w] < mn ? red2[w] : mn;
  }
  const long long span = mx + 1;
  const bool packed = can_pack(a.key_wrap32, mn, mx);  // the same verdict in every chunk: keys stay comparable

  const long long j = (long long)chunk * kChunk + tid;
  long long key = 0x7fffffffffffffffLL;  // padding of the last chunk sorts to its end
  int pay = (int)j;
  if (j < a.m) {
    int node, nbr;
    long long t, i;
    update_entry(a, j, node, nbr, t, i);
    key = update_key(node, t, span, a.key_wrap32);
    if (packed) key = packed_key(key, (int)j);
  }
  if (packed) bitonic_sort_one<true>(key, pay, s_key, s_pay, tid, kChunk);
  else bitonic_sort_one<false>(key, pay, s_key, s_pay, tid, kChunk);
  if constexpr (AGENT) {
    st_agent(&a.key[j], key);
    st_agent(&a.node[j], pay);
  } else {
    a.key[j] = key;
    a.node[j] = pay;  // chunk-sorted entry index
  }
}

// global rank of entry `tid` of chunk c = its rank in its chunk + sum over the other chunks of a binary search
// (pairs are unique, so the ranks are a permutation); k_all / p_all = the chunk-sorted (key, entry) arrays, in LDS
// (stand-alone kernel) or in global memory (riding along with a lookup launch: L2 hits, nobody waits for them).
// Chunk c keeps its len_c real entries first (padding sorted last), so [c*256, c*256 + len_c) is dense.
template <int GROUP>
__device__ __forceinline__ void update_merge_entry(const UpdateArgs& a, const long long* __restrict__ k_all,
                                                   const int* __restrict__ p_all, int c, int tid) {
  const int m = (int)a.m;
  const int e = c * kChunk + tid;
  if (e >= m) return;
  const long long key = k_all[e];
  const int pay = p_all[e];
  int rank = tid;
  const int chunks = (m + kChunk - 1) / kChunk;
  // GROUP binary searches advance together, so the dependent reads of one search hide behind the others' (9 steps each)
  constexpr int kMaxChunks = kBlockMaxM / kChunk;
  static_assert(kMaxChunks % GROUP == 0, "chunk groups");
#pragma unroll 1
  for (int g0 = 0; g0 < chunks; g0 += GROUP) {
    int lo[GROUP], hi[GROUP];
#pragma unroll
    for (int o = 0; o < GROUP; ++o) {
      const int oc = g0 + o;
      lo[o] = 0;
      const int len = (m - oc * kChunk) < kChunk ? (m - oc * kChunk) : kChunk;
      hi[o] = (oc < chunks && oc != c) ? len : 0;
    }
#pragma unroll 1
    for (int step = 0; step < 9; ++step) {  // 2^9 > kChunk
#pragma unroll
      for (int o = 0; o < GROUP; ++o) {
        if (lo[o] < hi[o]) {
          const int mid = (lo[o] + hi[o]) >> 1;
          if (pair_after(key, pay, k_all[(g0 + o) * kChunk + mid], p_all[(g0 + o) * kChunk + mid])) lo[o] = mid + 1;
          else hi[o] = mid;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < GROUP; ++o) rank += lo[o];
  }
  // everything the placement kernel would otherwise gather with dependent loads
  int node, nbr;
  long long t, i;
  update_entry(a, pay, node, nbr, t, i);
  const bool valid = node >= 0 && node < a.N && nbr >= 0 && nbr < a.N;
  if (!valid) atomicOr(a.status, TGMX_ST_EDGE_RANGE);
  Rec rec;
  rec.nbr = nbr;
  rec.eid = a.eid0 >= 0 ? (int)(a.eid0 + i) : -1;
  rec.ts = t;
  a.sorted_j[rank] = pay;
  a.sorted_node[rank] = valid ? node : -1;
  a.target[rank] = valid ? a.write_pos[node] % a.B : 0;  // write_pos only moves in the placement kernel
  a.sorted_rec[rank] = rec;
}

// The merge as it rides along with a lookup launch.  The lookups keep the memory pipeline full, so every DEPENDENT
// global read of a rider costs microseconds: the binary searches above (9 rounds per group of chunks) would outlast the
// launch.  Two levels instead: every 8th pair of every chunk is staged in LDS (one coalesced round, 6 KB at m = 4096),
// the search among a chunk's 32 samples runs in LDS, and only the last three steps (7 unknown pairs) read global
// memory -- for all other chunks at once, so three dependent rounds in all.  Chunks are padded to 256 pairs with keys
// that sort last (chunk sort), so no length bookkeeping is needed.
constexpr int kSampleEvery = 8, kSamplesPerChunk = kChunk / kSampleEvery;

struct SampleLds {
  long long key[(kBlockMaxM / kChunk) * kSamplesPerChunk];
  int pay[(kBlockMaxM / kChunk) * kSamplesPerChunk];
};

// NE entries per thread (entry `tid` of chunks c0 .. c0 + NE - 1), each searched in up to NC chunks; NE * NC
// searches advance together.  <1, 16>: one workgroup per chunk, any m <= 4096.  <4, 4>: ONE workgroup merges a whole
// batch of m <= 1024 entries (and goes on to place it, see below).
template <int NE, int NC, bool COH = false>  // COH: the placement follows in ANOTHER workgroup of this launch
__device__ __forceinline__ void update_merge_riding(const UpdateArgs& a, int c0, SampleLds& W) {
  const int m = (int)a.m;
  const int tid = threadIdx.x;
  const int chunks = (m + kChunk - 1) / kChunk;
  const long long* __restrict__ gk = a.key;
  const int* __restrict__ gp = a.node;
  for (int x = tid; x < chunks * kSamplesPerChunk; x += kChunk) {  // <= 2 rounds
    W.key[x] = ld_agent(&gk[x * kSampleEvery]);
    W.pay[x] = ld_agent(&gp[x * kSampleEvery]);
  }
  long long key[NE];
  int pay[NE];
  bool act[NE];
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    const int e = (c0 + q) * kChunk + tid;
    act[q] = e < m;
    key[q] = act[q] ? ld_agent(&gk[e]) : 0;
    pay[q] = act[q] ? ld_agent(&gp[e]) : 0;
  }
  __syncthreads();

  // pos[q][o] = number of pairs of chunk o known to sort before entry q; after the sample search it is
  // 8 * (#samples before mine - 1) + 1 (that sample is before mine, the next one is not), -1 when nothing of the chunk
  // is before mine (or there is no such search)
  int pos[NE][NC];
#pragma unroll
  for (int q = 0; q < NE; ++q) {
#pragma unroll
    for (int o = 0; o < NC; ++o) {
      int cnt = 0;  // samples of chunk o before mine, 0..32
      if (act[q] && o < chunks && o != c0 + q) {
        const long long* sk = W.key + o * kSamplesPerChunk;
        const int* sp = W.pay + o * kSamplesPerChunk;
        if (pair_after(key[q], pay[q], sk[kSamplesPerChunk - 1], sp[kSamplesPerChunk - 1])) {
          cnt = kSamplesPerChunk;
        } else {
#pragma unroll
          for (int st = kSamplesPerChunk / 2; st > 0; st >>= 1)
            if (pair_after(key[q], pay[q], sk[cnt + st - 1], sp[cnt + st - 1])) cnt += st;
        }
      }
      pos[q][o] = cnt > 0 ? (cnt - 1) * kSampleEvery + 1 : -1;
    }
  }
#pragma unroll
  for (int st = kSampleEvery / 2; st > 0; st >>= 1) {  // 3 dependent global rounds, every search in flight
    long long k2[NE][NC];
#pragma unroll
    for (int q = 0; q < NE; ++q)
#pragma unroll
      for (int o = 0; o < NC; ++o) k2[q][o] = pos[q][o] >= 0 ? ld_agent(&gk[o * kChunk + pos[q][o] + st - 1]) : 0;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
#pragma unroll
      for (int o = 0; o < NC; ++o) {
        if (pos[q][o] < 0) continue;
        bool after = key[q] > k2[q][o];
        if (key[q] == k2[q][o]) after = pay[q] > ld_agent(&gp[o * kChunk + pos[q][o] + st - 1]);  // equal unpacked keys only
        if (after) pos[q][o] += st;
      }
    }
  }
  // everything the placement would otherwise gather with dependent loads
  int node[NE], nbr[NE];
  long long t[NE], i[NE];
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    node[q] = nbr[q] = -1;
    t[q] = i[q] = 0;
    if (act[q]) update_entry(a, pay[q], node[q], nbr[q], t[q], i[q]);
  }
  int w[NE];
  bool valid[NE];
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    valid[q] = act[q] && node[q] >= 0 && node[q] < a.N && nbr[q] >= 0 && nbr[q] < a.N;
    w[q] = valid[q] ? a.write_pos[node[q]] : 0;  // write_pos only moves after the lookups
  }
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    if (!act[q]) continue;
    int rank = tid;
#pragma unroll
    for (int o = 0; o < NC; ++o) rank += pos[q][o] >= 0 ? pos[q][o] : 0;
    if (!valid[q]) atomicOr(a.status, TGMX_ST_EDGE_RANGE);
    Rec rec;
    rec.nbr = nbr[q];
    rec.eid = a.eid0 >= 0 ? (int)(a.eid0 + i[q]) : -1;
    rec.ts = t[q];
    if constexpr (COH) {
      st_agent(&a.sorted_j[rank], pay[q]);
      st_agent(&a.sorted_node[rank], valid[q] ? node[q] : -1);
      st_agent(&a.target[rank], w[q] % a.B);
    } else {
      a.sorted_j[rank] = pay[q];
      a.sorted_node[rank] = valid[q] ? node[q] : -1;
      a.target[rank] = w[q] % a.B;
    }
    a.sorted_rec[rank] = rec;  // (read by the commit launch only)
  }
}

// What rides along with which launch (tgmx_recency_step):
//   kSideSort  (hop 0): workgroup c chunk-sorts entries [256 c, 256 c + 256)
//   kSideMerge (hop 1, 1024 < m <= 4096): workgroup c ranks chunk c's entries; the placement is its own launch
//   kSideSortMerge / kSidePlaceOnly (hop 0 / hop 1 as two launches, m <= 1024, round 3): the chunk riders of hop 0 merge too (barrier between
//     them), hop 1's ONE rider only decides the placement -- review shape: hop 1 17.9 -> 7.9 us, step 29.0 -> 25.0 us
//   kSidePlace (hop 1, m <= 1024): ONE workgroup merges the whole batch and decides the placement (DEFER): after the
//              lookups a single launch commits records, write_pos and feature rows
//   kSideAll   (fused hop 0 + 1 launch, m <= 1024): ONE workgroup does all of it -- chunk sorts one after the other,
//              merge, placement decisions -- inside the single lookup launch
//   kSideSort / kSideMergePlace (hop 0 / hop 1 as two launches, m <= 1024, round 6): hop 0's riders only chunk-sort (no barrier inside a
//     launch: the launch boundary is the barrier), hop 1's riders rank their chunks and the last one out decides the placement -- the
//     update's chain is spread over BOTH lookup launches instead of pacing the first (review shape: 13.4 + 7.1 us -> see DESIGN 3.2)
//   kSideSortMerge (fused hop 0 + 1 launch, 1024 < m <= 4096): workgroup c chunk-sorts, all riders meet at a barrier of
//              their own (they are the launch's first <= 16 workgroups: resident together), then workgroup c merges
constexpr int kSideSort = 1, kSideMerge = 2, kSidePlace = 3, kSideAll = 4, kSideSortMerge = 5, kSidePlaceOnly = 6, kSideSortMergePlace = 7, kSideMergePlace = 8;

// Barrier between the `parts` rider workgroups of one launch: bar[0] counts arrivals, bar[1] is the generation.  It
// resets itself, so the words only have to be zero when the scratch buffer is first used.  What crosses it (the
// chunk-sorted pairs) is written and read with device-coherent accesses (st_agent / ld_agent), so the barrier itself is
// relaxed atomics: no L2 write-back or invalidate under the lookups' feet (with __threadfence() on both sides the
// fused launch took 42 us instead of 35 at m = 3200).  A barrier that never opens (non-zero scratch head) gives up
// after ~1 s and reports it.
__device__ __forceinline__ bool rider_barrier(int32_t* bar, int parts) {
  __syncthreads();  // every thread's sc1 stores have completed (workgroup release waits for them)
  bool ok = true;
  if (threadIdx.x == 0) {
    const int gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int prev = __hip_atomic_fetch_add(&bar[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == parts - 1) {
      __hip_atomic_store(&bar[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);  // the reset lands before the generation moves
      __hip_atomic_fetch_add(&bar[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      int spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 19)) {  // ~1 s of polling
          ok = false;
          break;
        }
      }
    }
  }
  __syncthreads();
  return ok;
}
constexpr int kRidePlaceMaxM = 1024;

// ---- tail commit -----------------------------------------------------------------------------------------------------
// bar[2] counts finished workgroups (lookups + rider), bar[3] finished tail workgroups; the last tail workgroup zeroes
// both, so the words only have to be zero when the scratch is first used (like the riders' barrier).
// Counting is two-level: 3000 same-address device-scope atomics serialise at the memory side (measured: the launch took
// 117 instead of 39 us with one counter), so workgroup b adds to group counter b % kTailGroups (own 64-byte lines:
// bar[16 + 16 g]) and only the workgroup that completes a group (it knows the group's size) moves the global word.
constexpr int kTailGroups = 32;
__device__ __forceinline__ void tail_signal(int32_t* bar, bool publish, unsigned bid, unsigned nblk) {
  __syncthreads();  // every thread's loads have returned and its stores have been issued and counted
  if (threadIdx.x == 0) {
    if (publish) {
      // the rider's placement decisions must be visible to tail workgroups on other XCDs (their L2s are not coherent
      // with this one): one write-back of this L2, once per launch; lookup workgroups publish nothing (relaxed)
      __threadfence();
      __hip_atomic_store(&bar[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // decisions readable
      __hip_atomic_fetch_add(&bar[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned g = bid % kTailGroups;
      const unsigned size = (nblk - g + kTailGroups - 1) / kTailGroups;  // workgroups b < nblk with b % kTailGroups == g
      int32_t* cnt = &bar[16 + 16 * g];
      const int prev = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == (int)size - 1) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        __hip_atomic_fetch_add(&bar[2], (int)size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// Workgroup `tb` of `tail_blocks`, positions [4 tb, 4 tb + 4), one wave each (the body of ring_update_feat_kernel<COMMIT>).
// Everything is READ before the launch-wide wait and only written after it: as soon as the rider has published its
// placement decisions (bar[4], set behind its write-back) the wave fetches its position's decision, record and feature
// row into registers; once every lookup workgroup has finished (bar[2] == need) what is left is stores.
__device__ __forceinline__ bool tail_wait(const int32_t* word, unsigned need) {
  int spins = 0;
  while ((unsigned)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(8);
    if (++spins > (1 << 22)) return false;  // seconds: the count can only be short if the scratch head was not zero at first use
  }
  return true;
}

__device__ __forceinline__ void tail_commit(const UpdateArgs& a, unsigned tb, unsigned tail_blocks, unsigned need) {
  __shared__ int ok_s;
  const int lane = lane_id();
  const long long p = (long long)tb * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (threadIdx.x == 0) ok_s = tail_wait(&a.barrier[4], 1u) ? 1 : 0;
  __syncthreads();
  bool ok = ok_s != 0;
  // ---- reads (the rider's scratch device-coherently: it was written through another XCD's L2) ----
  int row = -1, kept = 0, node = 0;
  long long rec_lo = 0, rec_hi = 0;
  constexpr int kMaxCols = 8;  // feature columns per lane held in registers: D <= 512, wider rows are copied after the wait
  float xv[kMaxCols];
  const float* x = nullptr;
  const bool wide = a.D > kMaxCols * kWave;
  if (ok && p < a.m) {
    row = ld_agent(&a.winner[p]);
    kept = ld_agent(&a.target[p]);
    if (row >= 0) {
      const long long* src = reinterpret_cast<const long long*>(&a.sorted_rec[p]);
      rec_lo = ld_agent(&src[0]);
      rec_hi = ld_agent(&src[1]);
    }
    if (kept > 0) node = ld_agent(&a.sorted_node[p]);
    if (row >= 0 && a.D > 0 && a.edge_x) {
      const long long j = ld_agent(&a.sorted_j[p]);
      x = a.edge_x + (j >= a.n ? j - a.n : j) * a.D;
      if (!wide) {
#pragma unroll
        for (int u = 0; u < kMaxCols; ++u) {
          const int c = lane + u * kWave;
          xv[u] = c < a.D ? x[c] : 0.f;
        }
      }
    }
  }
  // ---- every lookup workgroup and the rider done: nobody reads the rings any more ----
  if (threadIdx.x == 0) ok_s = (ok && tail_wait(&a.barrier[2], need)) ? 1 : 0;
  __syncthreads();
  ok = ok_s != 0;
  if (!ok) {
    if (threadIdx.x == 0) atomicOr(a.status, TGMX_ST_SCRATCH);
  } else if (p < a.m && !update_blocked(a)) {
    if (lane == 0) {
      if (row >= 0) {
        long long* dst = reinterpret_cast<long long*>(&a.ring[row]);
        dst[0] = rec_lo;
        dst[1] = rec_hi;
      }
      if (kept > 0) commit_write_pos(&a.write_pos[node], kept, a.B);
    }
    if (row >= 0 && a.D > 0) {
      float* __restrict__ o = a.ring_x + (long long)row * a.D;
      if (!x) {
        for (int c = lane; c < a.D; c += kWave) o[c] = 0.f;
      } else if (wide) {
        for (int c = lane; c < a.D; c += kWave) o[c] = x[c];
      } else {
#pragma unroll
        for (int u = 0; u < kMaxCols; ++u) {
          const int c = lane + u * kWave;
          if (c < a.D) o[c] = xv[u];
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(&a.barrier[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (int)tail_blocks - 1) {  // last one out: leave the words zero for the next launch
      __hip_atomic_store(&a.barrier[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.barrier[3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.barrier[4], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// PCAP: capacity of the riding placement (entries); 0 = this launch's riders only sort / merge.  The union is static LDS of EVERY
// workgroup of the launch, rider or not: 24.9 KB at 1024 entries caps a lookup launch at 6 workgroups per CU, 12.6 KB at 512
// (the single-rank wiki batch: m = 400) at 12, 6.2 KB without the placement tables at 25.
template <int PCAP>
union RiderLds {
  ChunkSortLds sort;
  SampleLds smp;
  PlaceLds<PCAP> place;
};
template <>
union RiderLds<0> {
  ChunkSortLds sort;
  SampleLds smp;
};

template <int PCAP = kRidePlaceMaxM>
__device__ __forceinline__ void update_side_work(const UpdateArgs& u, int stage, int block) {
  __shared__ RiderLds<PCAP> W;
  if (stage == kSideSort) {
    update_chunk_sort(u, block, W.sort);
  } else if (stage == kSideMerge) {
    update_merge_riding<1, kBlockMaxM / kChunk>(u, block, W.smp);
  } else if (stage == kSideSortMerge) {
    update_chunk_sort<true>(u, block, W.sort);
    if (!rider_barrier(u.barrier, (int)((u.m + kChunk - 1) / kChunk)) && threadIdx.x == 0) atomicOr(u.status, TGMX_ST_SCRATCH);
    update_merge_riding<1, kBlockMaxM / kChunk>(u, block, W.smp);
  } else if constexpr (PCAP > 0) {
    constexpr int NCH = PCAP / kChunk;  // chunks of a batch this rider takes whole: one entry of every chunk per thread
    if (stage == kSideSortMergePlace) {
      // fused hop 0 + 1 launch, 512 < m <= 1024 (a 2-rank share of the wiki batch: m = 800): one rider PER CHUNK sorts and, behind the
      // riders' barrier, ranks its chunk; the last one out decides the placement.  (One rider doing the four chunk sorts one after
      // the other, the merge and the placement took longer than the lookups around it: 42.7 us per step instead of 34.)
      const int chunks = (int)((u.m + kChunk - 1) / kChunk);
      __shared__ int last_out;
      update_chunk_sort<true>(u, block, W.sort);
      if (!rider_barrier(u.barrier, chunks) && threadIdx.x == 0) atomicOr(u.status, TGMX_ST_SCRATCH);
      update_merge_riding<1, NCH, true>(u, block, W.smp);
      __syncthreads();  // every thread's device-coherent stores have completed
      if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(&u.barrier[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_out = prev == chunks - 1;
        if (last_out) __hip_atomic_store(&u.barrier[5], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
      }
      __syncthreads();
      if (!last_out) return;
      SortLds<1> none;
      update_block_body<NCH, PCAP, true, true, true>(u, W.place, none);
      return;
    }
    if (stage == kSideMergePlace) {
      // the chunk sorts ran in the PREVIOUS launch (hop 0, kSideSort): rank this rider's chunk; the last rider out decides the placement
      const int chunks = (int)((u.m + kChunk - 1) / kChunk);
      __shared__ int last_mp;
      update_merge_riding<1, NCH, true>(u, block, W.smp);
      __syncthreads();  // every thread's device-coherent stores have completed
      if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(&u.barrier[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_mp = prev == chunks - 1;
        if (last_mp) __hip_atomic_store(&u.barrier[5], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
      }
      __syncthreads();
      if (!last_mp) return;
      SortLds<1> none;
      update_block_body<NCH, PCAP, true, true, true>(u, W.place, none);
      return;
    }
    if (stage == kSideAll) {
      const int chunks = (int)((u.m + kChunk - 1) / kChunk);
#pragma unroll 1
      for (int c = 0; c < NCH; ++c) {
        if (c < chunks) update_chunk_sort(u, c, W.sort);
        __syncthreads();  // W.sort is reused; the chunk-sorted pairs written above are read below by other threads
      }
    }
    if (stage != kSidePlaceOnly) {  // (kSidePlaceOnly: the hop-0 launch's riders merged already -- kSideSortMerge)
      update_merge_riding<NCH, NCH>(u, 0, W.smp);
      __syncthreads();  // the sorted arrays written above are read below by other threads of this workgroup
    }
    SortLds<1> none;
    update_block_body<NCH, PCAP, true, true>(u, W.place, none);
  }
}


// wave_prefix_count for a GROUP of GL lanes (64 / GL groups per wave, each with its own [a, z)): GL probes per round.
// The loop runs while ANY group of the wave still has more than GL candidates (finished groups idle through it).
template <int GL>
__device__ __forceinline__ long long group_prefix_count(const Rec* recs, long long a, long long z, long long bound, int gl, int sub) {
  constexpr unsigned long long kMask = GL == 64 ? ~0ull : ((1ull << GL) - 1);
  long long lo = a, hi = z;
  while (__any(hi - lo > GL)) {
    const bool busy = hi - lo > GL;
    const long long len = hi - lo;
    const long long stride = (len + GL - 1) / GL;
    long long idx = lo + (long long)(gl + 1) * stride - 1;
    if (idx > hi - 1) idx = hi - 1;
    const bool less = busy && (long long)recs[idx].eid < bound;
    const int c = __popcll((__ballot(less) >> (sub * GL)) & kMask);
    if (busy) {
      long long nlo = lo, nhi = hi;
      if (c > 0) {
        long long p = lo + (long long)c * stride - 1;
        if (p > hi - 1) p = hi - 1;
        nlo = p + 1;
      }
      if (c < GL) {
        long long p = lo + (long long)(c + 1) * stride - 1;
        if (p > hi - 1) p = hi - 1;
        nhi = p;  // recs[p] is >= bound: the answer is <= p
      }
      lo = nlo;
      hi = nhi < nlo ? nlo : nhi;
    }
  }
  const long long idx = lo + gl;
  const bool less = idx < hi && (long long)recs[idx].eid < bound;
  return lo + __popcll((__ballot(less) >> (sub * GL)) & kMask) - a;
}

// ---- one seed's lookup, in pieces (shared by the per-hop kernel and the fused hop-0 + hop-1 kernel) -----------------
// hop-0 seed s: from the seed groups (publishing the concatenated arrays when `publish`) or the plain arrays
__device__ __forceinline__ void fetch_seed(const LookupArgs& a, long long s, int lane, bool publish, int& n, long long& q) {
  if (a.grp.groups > 0) {
    const int32_t* pn = a.grp.nid[0];
    const int64_t* pt = a.grp.ts[0];
    long long base = 0;
    int sel = 0;
#pragma unroll
    for (int g = 1; g < TGMX_MAX_SEED_GROUPS; ++g) {
      if (g < a.grp.groups && s >= a.grp.end[g - 1]) {
        pn = a.grp.nid[g];
        pt = a.grp.ts[g];
        base = a.grp.end[g - 1];
        sel = g;
      }
    }
    const bool gen = sel == a.grp.gen;  // wave-uniform
    n = gen ? negative_draw(a.grp.gen_seed, a.grp.gen_call, a.grp.gen_index0 + (unsigned long long)(s - base), a.grp.gen_low, a.grp.gen_range) : pn[s - base];
    q = pt[s - base];
    if (publish && lane == 0) {
      a.grp.out_nid[s] = n;
      a.grp.out_ts[s] = q;
      if (gen) {
        a.grp.gen_out[s - base] = n;
        a.grp.gen_out_ts[s - base] = q;
      }
    }
  } else {
    n = a.seeds[s];
    q = a.qtimes[s];
  }
}

__device__ __forceinline__ void check_seed(const LookupArgs& a, int n, long long q, int allow_pad, int lane) {
  if (lane == 0) {
    int st = 0;
    if (n >= a.N || n < -1 || (n == -1 && !allow_pad)) st |= TGMX_ST_SEED_RANGE;
    if (q < 0 && !allow_pad) st |= TGMX_ST_SEED_TIME;
    if (st) atomicOr(a.status, st);
  }
}

// window [w0, w0 + wlen) of node n in "oldest -> newest" order
struct Window {
  long long w0;  // CSR: absolute record index; RING: row base
  int wlen, wrot;
};

template <bool RING>
__device__ __forceinline__ Window find_window(const LookupArgs& a, int n, bool live, int lane) {
  Window w;
  w.w0 = 0;
  w.wlen = 0;
  w.wrot = 0;
  if (live) {
    if (RING) {
      w.w0 = (long long)n * a.B;
      w.wrot = a.write_pos[n] % a.B;  // unrolled position i lives in slot (wrot + i) % B
      w.wlen = a.B;
    } else {
      const long long ra = a.indptr[n], rz = a.indptr[n + 1];
      const long long p_hi = ra + wave_prefix_count(a.recs, ra, rz, a.ev_hi, lane);
      const long long p_lo = a.ev_lo <= 0 ? ra : ra + wave_prefix_count(a.recs, ra, rz, a.ev_lo, lane);
      w.w0 = p_hi - a.B > p_lo ? p_hi - a.B : p_lo;
      w.wlen = (int)(p_hi - w.w0);
    }
  }
  return w;
}

template <bool RING>
__device__ __forceinline__ long long slot_of(const Window& w, int B, int i) {
  if (RING) {
    int sl = w.wrot + i;
    if (sl >= B) sl -= B;
    return w.w0 + sl;
  }
  return w.w0 + i;
}

// B, k <= 64: the whole window in one wave.  Lane c < k gets output slot c of the k-wide row: (has, nbr, ts) and the
// feature row it gathers from (ring slot / edge id; -1 for a pad)
struct SmallPick {
  bool has;
  int nbr, src, eid;
  long long ts;
};

template <bool RING>
__device__ __forceinline__ SmallPick small_pick(const LookupArgs& a, int n, long long q, int k, bool live, int lane) {
  const int B = a.B;
  const Window w = find_window<RING>(a, n, live, lane);
  Rec r;
  r.nbr = -1; r.eid = 0; r.ts = 0;
  if (RING) {
    // the row is read in slot order (no wait for write_pos) and rotated into time order by a shuffle
    if (live && lane < B) r = a.recs[(long long)n * B + lane];
    int from_slot = w.wrot + lane;
    if (from_slot >= B) from_slot -= B;
    if (lane >= B) from_slot = lane;
    r.nbr = __shfl(r.nbr, from_slot);
    r.ts = __shfl(r.ts, from_slot);
    if (a.out_eid || a.out_eid1) r.eid = __shfl(r.eid, from_slot);  // (only a caller that asked for edge ids reads it: wave-uniform)
  } else if (lane < w.wlen) {
    r = a.recs[slot_of<RING>(w, B, lane)];
  }
  const bool ok = lane < w.wlen && r.nbr >= 0 && r.ts < q;
  const unsigned long long m = __ballot(ok);
  const int cnt = m ? 64 - __clzll((long long)m) : 0;  // 1 + unrolled position of the rightmost entry with ts < q
  const int i = cnt - k + lane;                        // unrolled position feeding output slot `lane`
  const int from = i > 0 ? i : 0;
  const int g_nbr = __shfl(r.nbr, from);
  const int g_eid = __shfl(r.eid, from);
  const long long g_ts = __shfl(r.ts, from);
  SmallPick o;
  o.has = i >= 0 && g_nbr >= 0;
  o.nbr = o.has ? g_nbr : -1;
  o.ts = o.has ? g_ts : 0;
  o.src = o.has ? ((RING || a.x_by_pos) ? (int)slot_of<RING>(w, B, from) : g_eid) : -1;
  o.eid = o.has ? g_eid : -1;
  return o;
}

// phase B: stream the [k, D] block of row s; lds_eid[c] = feature row of output slot c (-1: zeros)
// Round 3: straight-line trips -- every LDS lookup of the trip first, then U UNCONDITIONAL loads (a piece of a pad slot reads row 0 and
// is replaced by zeros), then the stores: a branch around each load put an LDS round trip in front of every one of them, and four in
// flight per lane were two trips more per seed than eight.
template <int VEC>
__device__ __forceinline__ void gather_rows(const LookupArgs& a, long long s, int k, int lane, const int* lds_eid, float* out_x, int first_slot) {
  using V = typename VecOf<VEC>::type;
  const V* __restrict__ X = reinterpret_cast<const V*>(a.edge_x);
  V* __restrict__ O = reinterpret_cast<V*>(out_x + s * (long long)k * a.D);
  const int total = k * a.row_vecs;
  constexpr int U = 8;
  for (int f0 = first_slot * a.row_vecs + lane; f0 < total; f0 += kWave * U) {
    long long idx[U];
    bool has[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u * kWave;
      const int fc = f < total ? f : f0;  // (f0 < total: a piece of this very row)
      const int slot = (int)a.dv.div_nb((uint32_t)fc);
      const int col = fc - slot * a.row_vecs;
      const int e = lds_eid[slot];
      has[u] = e >= 0;
      idx[u] = has[u] ? (long long)e * a.row_vecs + col : 0;
    }
    V v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = X[idx[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u * kWave;
      if (f < total) O[f] = has[u] ? v[u] : zero_vec<V>();
    }
  }
}

// the k most recent neighbors of (n, q) into row s of (out_nid, out_ts, out_x)
template <bool RING, int VEC, bool SMALL>
__device__ __forceinline__ void lookup_seed(const LookupArgs& a, long long s, int n, long long q, int k, int lane, int* lds_eid,
                                            int32_t* out_nid, int64_t* out_ts, float* out_x, int32_t* out_valid, int32_t* out_valid_prev,
                                            int32_t* out_eid = nullptr) {
  const bool live = n >= 0 && n < a.N;
  // the row's SPAN: slots from its leftmost non-pad one to the end (0: all pads).  Valid slots sit at the right end of a row, but a
  // ring can hold a pad record between real ones (the oracle's lookup keeps it: an interior -1), so it is the leftmost
  // non-pad slot that bounds what has to be written, not the count.  Wave-uniform.
  int v_new = 0;
  if (SMALL) {
    const SmallPick o = small_pick<RING>(a, n, q, k, live, lane);
    if (lane < k) {
      out_nid[s * k + lane] = o.nbr;
      out_ts[s * k + lane] = o.ts;
      if (out_eid) out_eid[s * k + lane] = o.eid;
      lds_eid[lane] = o.src;
    }
    const unsigned long long m = __ballot(lane < k && o.has);
    v_new = m ? k - __builtin_ctzll(m) : 0;
  } else {
    const int B = a.B;
    const Window w = find_window<RING>(a, n, live, lane);
    int cnt = 0;  // 1 + unrolled position of the rightmost entry with ts < q
    for (int top = w.wlen; top > 0 && cnt == 0; top -= kWave) {
      const int lo = top > kWave ? top - kWave : 0;
      const int i = lo + lane;
      bool ok = false;
      if (i < top) {
        const Rec r = a.recs[slot_of<RING>(w, B, i)];
        ok = r.nbr >= 0 && r.ts < q;
      }
      const unsigned long long m = __ballot(ok);
      if (m) cnt = lo + 64 - __clzll((long long)m);
    }
    for (int c = lane; c < k; c += kWave) {
      const int i = cnt - k + c;
      Rec r;
      r.nbr = -1; r.eid = 0; r.ts = 0;
      if (i >= 0) r = a.recs[slot_of<RING>(w, B, i)];
      const bool has = r.nbr >= 0;
      out_nid[s * k + c] = has ? r.nbr : -1;
      out_ts[s * k + c] = has ? r.ts : 0;
      if (out_eid) out_eid[s * k + c] = has ? r.eid : -1;
      lds_eid[c] = has ? ((RING || a.x_by_pos) ? (int)slot_of<RING>(w, B, i) : r.eid) : -1;
      if (has && k - c > v_new) v_new = k - c;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const int other = __shfl_xor(v_new, o);
      v_new = other > v_new ? other : v_new;
    }
  }
  if (a.D == 0 || out_x == nullptr) return;  // no features, or the consumer gathers them by edge id (out_eid)
  int first_slot = 0;
  if (out_valid) {  // the row's previous contents are known: leave the zeros left of both valid tails alone
    const int v_old = out_valid[s];
    first_slot = k - (v_old > v_new ? v_old : v_new);
    if (lane == 0) {
      out_valid[s] = v_new;
      if (out_valid_prev) out_valid_prev[s] = v_old;
    }
  }
  __builtin_amdgcn_wave_barrier();  // lds_eid written above is read cross-lane below
  gather_rows<VEC>(a, s, k, lane, lds_eid, out_x, first_slot);
  __builtin_amdgcn_wave_barrier();  // the next seed reuses lds_eid
}

template <bool RING, int VEC, bool SMALL, bool RIDE>
__global__ __launch_bounds__(256) void recency_lookup_kernel(const LookupArgs a, const UpdateArgs u) {
  extern __shared__ __attribute__((aligned(16))) int lds_eid_all[];
  unsigned bid = blockIdx.x, nblk = gridDim.x;
  if constexpr (RING && RIDE) {
    if (bid < a.side_blocks) {
      update_side_work(u, a.side_stage, (int)bid);
      if (a.tail_blocks) tail_signal(u.barrier, true, 0, 0);
      return;
    }
    bid -= a.side_blocks;
    nblk -= a.side_blocks + a.tail_blocks;
    if (bid >= nblk) {  // the launch's last workgroups: the update's write-back, once everyone else is done
      tail_commit(u, bid - nblk, a.tail_blocks, nblk + a.side_blocks);
      return;
    }
  }
  const int lane = lane_id();
  const int wave_in_block = threadIdx.x >> 6;
  int* lds_eid = lds_eid_all + wave_in_block * a.k;
  const long long waves_total = (long long)nblk * (blockDim.x >> 6);
  for (long long s = (long long)bid * (blockDim.x >> 6) + wave_in_block; s < a.S; s += waves_total) {
    int n;
    long long q;
    fetch_seed(a, s, lane, true, n, q);
    check_seed(a, n, q, a.allow_pad, lane);
    lookup_seed<RING, VEC, SMALL>(a, s, n, q, a.k, lane, lds_eid, a.out_nid, a.out_ts, a.out_x, a.out_valid, a.out_valid_prev, a.out_eid);
  }
  if constexpr (RING && RIDE) {
    if (a.tail_blocks) tail_signal(u.barrier, false, bid, nblk);
  }
}

// Hop 0 and hop 1 in ONE launch (B, k0, k1 <= 64).  Hop 1's seeds are hop 0's outputs, but the rings do not move
// between the two, so the wave of hop-1 row (s0, j) re-derives its seed itself: it repeats hop 0's window search for
// seed s0 (one 16-byte read per lane of a row that 1 + k0 waves share: L2 hits) and takes output slot j.  That removes
// the dependency between the launches: the few hundred hop-0 seeds -- a launch bound by its three dependent reads, not
// by bandwidth -- run inside the big hop-1 launch instead of in front of it.  Waves [0, S0) are hop 0 (they also
// publish the concatenated seeds), waves [S0, S0 + S0 k0) are hop 1; results are identical to the two launches.
template <bool RING, int VEC, int PCAP = kRidePlaceMaxM>
__global__ __launch_bounds__(256) void recency_lookup_fused01_kernel(const LookupArgs a, const UpdateArgs u) {
  extern __shared__ __attribute__((aligned(16))) int lds_eid_all[];
  unsigned bid = blockIdx.x, nblk = gridDim.x;
  if constexpr (RING) {
    if (bid < a.side_blocks) {
      update_side_work<PCAP>(u, a.side_stage, (int)bid);
      if (a.tail_blocks) tail_signal(u.barrier, true, 0, 0);
      return;
    }
    bid -= a.side_blocks;
    nblk -= a.side_blocks + a.tail_blocks;
    if (bid >= nblk) {  // the launch's last workgroups: the update's write-back, once everyone else is done
      tail_commit(u, bid - nblk, a.tail_blocks, nblk + a.side_blocks);
      return;
    }
  }
  const int lane = lane_id();
  const int wave_in_block = threadIdx.x >> 6;
  const int k0 = a.k, k1 = a.k1;
  int* lds_eid = lds_eid_all + wave_in_block * (k0 > k1 ? k0 : k1);
  const long long waves_total = (long long)nblk * (blockDim.x >> 6);
  const long long S0 = a.S, S1 = a.S * k0;
  for (long long w = (long long)bid * (blockDim.x >> 6) + wave_in_block; w < S0 + S1; w += waves_total) {
    int n;
    long long q;
    if (w < S0) {
      fetch_seed(a, w, lane, true, n, q);
      check_seed(a, n, q, 0, lane);
      lookup_seed<RING, VEC, true>(a, w, n, q, k0, lane, lds_eid, a.out_nid, a.out_ts, a.out_x, a.out_valid, a.out_valid_prev, a.out_eid);
    } else {
      const long long idx = w - S0;
      const long long s0 = idx / k0;
      const int j = (int)(idx - s0 * k0);
      int n0;
      long long q0;
      fetch_seed(a, s0, lane, false, n0, q0);
      const SmallPick o = small_pick<RING>(a, n0, q0, k0, n0 >= 0 && n0 < a.N, lane);
      n = __shfl(o.nbr, j);
      q = __shfl(o.ts, j);
      lookup_seed<RING, VEC, true>(a, idx, n, q, k1, lane, lds_eid, a.out_nid1, a.out_ts1, a.out_x1, a.out_valid1, a.out_valid_prev1, a.out_eid1);
    }
  }
  if constexpr (RING) {
    if (a.tail_blocks) tail_signal(u.barrier, false, bid, nblk);
  }
}

// Narrow feature rows (k * D small, e.g. D = 16): one wave per seed moves ~1 KB behind a chain of dependent loads, so
// the launch is latency-bound.  The packed variant gives every seed a GROUP of GL lanes (B, k <= GL): 64 / GL seeds per
// wave, the same ballot / shuffle logic inside the group's slice of the wave, 2-4x the loads in flight.  Plain seed
// arrays only (hops >= 1).  RING: streaming rings (and the riders of the ring update); else the static index, with
// the batch-boundary prefix searches done per group.
// ---- narrow rows: one GROUP of GL lanes per seed (64 / GL seeds per wave) ----------------------------------------------
// the k most recent neighbors of (n, q) inside the group's slice of the wave: lane gl < k gets output slot gl
struct GroupPick {
  bool has;
  int nbr, src;  // src: feature row of the slot (ring slot / edge id), -1 for a pad
  long long ts;
};

template <bool RING, int GL>
__device__ __forceinline__ GroupPick group_pick(const LookupArgs& a, int n, long long q, int k, bool live, int gl, int sub) {
  const int B = a.B;
  // window of <= B records in time order: the ring row rotated by write_pos, or the last B visible index entries
  long long w0 = 0;
  int wrot = 0, wlen = 0;
  if constexpr (RING) {
    w0 = (long long)(live ? n : 0) * B;
    wrot = live ? a.write_pos[n] % B : 0;
    wlen = live ? B : 0;
  } else {
    const long long ra = live ? a.indptr[n] : 0, rz = live ? a.indptr[n + 1] : 0;
    const long long p_hi = ra + group_prefix_count<GL>(a.recs, ra, rz, a.ev_hi, gl, sub);
    const long long p_lo = a.ev_lo <= 0 ? ra : ra + group_prefix_count<GL>(a.recs, ra, rz, a.ev_lo, gl, sub);
    w0 = p_hi - B > p_lo ? p_hi - B : p_lo;
    wlen = (int)(p_hi - w0);
  }
  Rec r;
  r.nbr = -1; r.eid = 0; r.ts = 0;
  if (gl < wlen) r = a.recs[w0 + gl];  // RING: slot order (no wait for write_pos), rotated into time order below
  if constexpr (RING) {
    int from_slot = wrot + gl;
    if (from_slot >= B) from_slot -= B;
    if (gl >= B) from_slot = gl;
    r.nbr = __shfl(r.nbr, sub * GL + from_slot);
    r.ts = __shfl(r.ts, sub * GL + from_slot);
  }
  const bool ok = gl < wlen && r.nbr >= 0 && r.ts < q;
  const unsigned long long m = (__ballot(ok) >> (sub * GL)) & ((GL == 64) ? ~0ull : ((1ull << GL) - 1));
  const int cnt = m ? 64 - __clzll((long long)m) : 0;  // 1 + position of the rightmost valid entry, inside the group
  const int i = cnt - k + gl;
  const int from = i > 0 ? i : 0;
  const int g_nbr = __shfl(r.nbr, sub * GL + from);
  const int g_eid = __shfl(r.eid, sub * GL + from);
  const long long g_ts = __shfl(r.ts, sub * GL + from);
  GroupPick o;
  o.has = i >= 0 && g_nbr >= 0;
  o.nbr = o.has ? g_nbr : -1;
  o.ts = o.has ? g_ts : 0;
  int sl = wrot + from;
  if (sl >= B) sl -= B;
  o.src = o.has ? ((RING || a.x_by_pos) ? (int)(w0 + sl) : g_eid) : -1;
  return o;
}

// row s of (out_nid, out_ts, out_x) from the group's pick; k is wave-uniform
template <int VEC, int GL>
__device__ __forceinline__ void group_emit(const LookupArgs& a, bool act, long long s, int k, const GroupPick& o, int* lds_eid, int gl, int sub,
                                           int32_t* out_nid, int64_t* out_ts, float* out_x, int32_t* out_valid, int32_t* out_valid_prev) {
  using V = typename VecOf<VEC>::type;
  if (act && gl < k) {
    out_nid[s * k + gl] = o.nbr;
    out_ts[s * k + gl] = o.ts;
    lds_eid[gl] = o.src;
  }
  if (a.D == 0) return;
  int first_slot = 0;
  if (out_valid) {  // delta feature writes: see LookupArgs::out_valid
    const unsigned long long m = (__ballot(act && gl < k && o.has) >> (sub * GL)) & ((GL == 64) ? ~0ull : ((1ull << GL) - 1));
    const int v_new = m ? k - __builtin_ctzll(m) : 0;  // span from the leftmost non-pad slot (see lookup_seed)
    int v_old = 0;
    if (act) {
      v_old = out_valid[s];
      first_slot = k - (v_old > v_new ? v_old : v_new);
    }
    __builtin_amdgcn_wave_barrier();  // every lane of the group has read the old span
    if (act && gl == 0) {
      out_valid[s] = v_new;
      if (out_valid_prev) out_valid_prev[s] = v_old;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (act) {
    const V* __restrict__ X = reinterpret_cast<const V*>(a.edge_x);
    V* __restrict__ O = reinterpret_cast<V*>(out_x + s * (long long)k * a.D);
    const int total = k * a.row_vecs;
    constexpr int U = 4;
    for (int f0 = first_slot * a.row_vecs + gl; f0 < total; f0 += GL * U) {
      V v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + u * GL;
        v[u] = zero_vec<V>();
        if (f < total) {
          const int slot = (int)a.dv.div((uint32_t)f);
          const int col = f - slot * a.row_vecs;
          const int e = lds_eid[slot];
          if (e >= 0) v[u] = X[(long long)e * a.row_vecs + col];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + u * GL;
        if (f < total) O[f] = v[u];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// RIDE = false: no rider workgroups in this launch (large batches take the radix-sort update on the side stream, the
// static index has no update) -- the instantiation then carries no rider LDS (24.9 KB static, i.e. 6 workgroups per CU)
// and the launch runs at twice the occupancy, which is what this latency-bound kernel lives on.
// (Hop 0 inside this launch -- hop-1 groups re-deriving their seed like recency_lookup_fused01_kernel does for wide rows --
// was measured and dropped: the re-derivation doubles the chain of dependent reads this kernel is bound by; review shape
// 18.0 + 6.4 us as two launches vs 32.6 us fused, comment shape 123 + 18 vs 186 us.)
template <bool RING, int VEC, int GL, bool RIDE>
__global__ __launch_bounds__(256) void lookup_packed_kernel(const LookupArgs a, const UpdateArgs u) {
  constexpr int kGroups = kWave / GL;
  extern __shared__ __attribute__((aligned(16))) int lds_eid_all[];
  unsigned bid = blockIdx.x, nblk = gridDim.x;
  if constexpr (RING && RIDE) {
    if (bid < a.side_blocks) {
      update_side_work(u, a.side_stage, (int)bid);
      if (a.tail_blocks) tail_signal(u.barrier, true, 0, 0);
      return;
    }
    bid -= a.side_blocks;
    nblk -= a.side_blocks + a.tail_blocks;
    if (bid >= nblk) {  // the launch's last workgroups: the update's write-back, once everyone else is done
      tail_commit(u, bid - nblk, a.tail_blocks, nblk + a.side_blocks);
      return;
    }
  }
  const int lane = lane_id();
  const int sub = lane / GL, gl = lane - sub * GL;
  const int wave_in_block = threadIdx.x >> 6;
  const int k = a.k;
  int* lds_eid = lds_eid_all + (wave_in_block * kGroups + sub) * k;
  const long long waves_total = (long long)nblk * (blockDim.x >> 6);
  const long long n_rounds = (a.S + kGroups - 1) / kGroups;
  for (long long w = (long long)bid * (blockDim.x >> 6) + wave_in_block; w < n_rounds; w += waves_total) {
    const long long s = w * kGroups + sub;
    const bool act = s < a.S;
    int n = -1;
    long long q = 0;
    if (act) {
      if (a.grp.groups > 0) {  // hop 0: the seed comes from its group (or is drawn) and the group's first lane publishes it
        fetch_seed(a, s, gl, true, n, q);
      } else {
        n = a.seeds[s];
        q = a.qtimes[s];
      }
    }
    if (act && gl == 0) {
      int st = 0;
      if (n >= a.N || n < -1 || (n == -1 && !a.allow_pad)) st |= TGMX_ST_SEED_RANGE;
      if (q < 0 && !a.allow_pad) st |= TGMX_ST_SEED_TIME;
      if (st) atomicOr(a.status, st);
    }
    const GroupPick o = group_pick<RING, GL>(a, n, q, k, n >= 0 && n < a.N, gl, sub);
    group_emit<VEC, GL>(a, act, s, k, o, lds_eid, gl, sub, a.out_nid, a.out_ts, a.out_x, a.out_valid, a.out_valid_prev);
  }
  if constexpr (RING && RIDE) {
    if (a.tail_blocks) tail_signal(u.barrier, false, bid, nblk);
  }
}


// ---- narrow rows, a TILE of 64 seeds per wave: one LANE per seed for the index work, the whole wave for the copies -------------
// The packed kernel above keeps 2-4 seeds in flight per wave behind a chain of dependent reads (seed -> write_pos / window ->
// feature rows -> stores); counters say its waves spend 82 % of their life waiting at memory latencies that are close to
// unloaded (TCP -> TCC 780 cycles, TCC -> fabric 1150), i.e. the launch is bound by how few requests a CU has in flight, not by
// bandwidth (ablation: with every memory access but the seed reads removed it still takes 40 of its 139 us).  Here a wave
// takes 64 consecutive seeds:
//   index phase  lane = seed: its window of <= BCAP 16-byte records arrives as BCAP independent loads per lane (64 x BCAP
//                in flight per wave); the pick is straight-line code over registers; ids / times / feature-row sources of the
//                tile are staged in LDS in the outputs' own row-major layout
//   flush        the tile's [64, k] ids and times leave as flat, fully coalesced 16-byte stores
//   copy phase   the tile's [64, k, D] block is ONE contiguous stretch of the output: the wave streams it as flat 16-byte pieces,
//                8 independent loads in flight per lane, sources looked up in LDS; pieces left of a row's first changing slot
//                (delta writes) are skipped
// CSR: the batch-boundary prefix searches are per-lane 4-ary searches (64 independent searches per wave).
// Stores with a per-lane predicate and NO branch: a buffer store whose offset lies outside the descriptor's range is dropped by
// the hardware.  (A branch around every store makes hipcc's wait-count insertion give up counting -- it then drains the whole
// queue, `s_waitcnt vmcnt(0)`, before EVERY store, i.e. each store waits for the one before it to reach L2: the first version of
// this copy loop ran 8 stores per batch as 8 serial round trips.)
typedef unsigned int tgmx_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int tgmx_u2 __attribute__((ext_vector_type(2)));
// (off: per-lane byte offset, range-checked; soff: wave-uniform byte offset added after the check)
__device__ __forceinline__ void buffer_store_vec(float4 v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<tgmx_u4*>(&v), r, off, soff, 0);
}
__device__ __forceinline__ void buffer_store_vec(float2 v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<tgmx_u2*>(&v), r, off, soff, 0);
}
__device__ __forceinline__ void buffer_store_vec(float v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, soff, 0);
}
// ... and loads whose out-of-range lanes read zeros
__device__ __forceinline__ void buffer_load_vec(float4& v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  const tgmx_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  v = *reinterpret_cast<const float4*>(&x);
}
__device__ __forceinline__ void buffer_load_vec(float2& v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  const tgmx_u2 x = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
  v = *reinterpret_cast<const float2*>(&x);
}
__device__ __forceinline__ void buffer_load_vec(float& v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
template <typename T>
__device__ __forceinline__ T* wave_uniform_ptr(T* p) {  // tell the compiler what we know: the value is the same in every lane
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// The tile's feature rows: Lsrc[seed * k + slot] = the row each output slot is copied from (-1: zeros), Llist[0 .. n_rows) = the
// output slots that have to be written at all, as (seed << 8 | slot) -- the rows of pad seeds and, with delta writes, everything left
// of a row's first changing slot are not in it (47 % of a comment-shaped tile: two trips of the loop below instead of five walking
// every slot with most lanes predicated off).
// SMALL_TABLE: the feature table is smaller than 4 GB (the launch checks: rings of up to ~16 M rows of 64 B): it sits behind a buffer
// descriptor too, a piece's address is ONE register, and a piece without a source simply reads out of range (zeros) -- no select.
template <int VEC, bool SMALL_TABLE>
__device__ __forceinline__ void tile_copy(const LookupArgs& a, long long t, int rows, int k, int lane, const int* Lsrc,
                                          const unsigned short* Llist, int n_rows, long long table_rows) {
  using V = typename VecOf<VEC>::type;
  const V* __restrict__ X = reinterpret_cast<const V*>(a.edge_x);
  const int rv = a.row_vecs, total = n_rows * rv;
  // the tile's [rows, k, D] block of the output behind one wave-uniform descriptor
  float* base = wave_uniform_ptr(a.out_x + t * 64 * (long long)k * a.D);
  const __amdgpu_buffer_rsrc_t O = __builtin_amdgcn_make_buffer_rsrc(base, 0, rows * k * rv * (int)sizeof(V), 0x00020000);
  const __amdgpu_buffer_rsrc_t T = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.edge_x), 0,
                                                                     SMALL_TABLE ? (int)(unsigned)(table_rows * rv * (long long)sizeof(V)) : 0, 0x00020000);
  constexpr int U = SMALL_TABLE ? 16 : 12;  // pieces per lane and trip: U independent 16-byte loads in flight per lane
  constexpr unsigned kDrop = 0xFFFFFFF0u;  // outside every tile and every table: the store is dropped, the load reads zeros
  // vmcnt(0), once, in front of the loop: with memory operations of the phases before still in hipcc's books at loop entry, its
  // wait-count insertion puts a full drain at the TOP of every trip instead (every trip then pays a store round trip)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // One trip: every LDS lookup, then U unconditional loads, then U range-checked stores -- straight-line code.
  for (int fb = 0; fb < total; fb += kWave * U) {  // wave-uniform trip count
    unsigned off[U];
    V v[U];
    if constexpr (SMALL_TABLE) {
      unsigned src[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = fb + u * kWave + lane;
        const bool in = f < total;
        const int fc = in ? f : 0;
        const int row = (int)a.dv.div_nb((uint32_t)fc);
        const int col = fc - row * rv;
        const int entry = Llist[row];
        const int at = (entry >> 8) * k + (entry & 255);  // seed * k + slot
        const int e = Lsrc[at];
        src[u] = (in & (e >= 0)) ? ((unsigned)e * (unsigned)rv + (unsigned)col) * (unsigned)sizeof(V) : kDrop;
        off[u] = in ? (unsigned)(at * rv + col) * (unsigned)sizeof(V) : kDrop;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) buffer_load_vec(v[u], T, src[u]);
    } else {
      long long idx[U];
      bool has[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = fb + u * kWave + lane;
        const bool in = f < total;
        const int fc = in ? f : 0;
        const int row = (int)a.dv.div_nb((uint32_t)fc);
        const int col = fc - row * rv;
        const int entry = Llist[row];
        const int at = (entry >> 8) * k + (entry & 255);
        const int e = Lsrc[at];
        has[u] = in & (e >= 0);
        idx[u] = has[u] ? (long long)e * rv + col : 0;  // a piece without a source reads row 0 and is replaced by zeros
        off[u] = in ? (unsigned)(at * rv + col) * (unsigned)sizeof(V) : kDrop;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = X[idx[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = has[u] ? v[u] : zero_vec<V>();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) buffer_store_vec(v[u], O, off[u], 0u);
  }
}

// records of [a, z) with eid < bound form a prefix: their number, by a 4-ary search private to the lane (3 probes in flight per round)
__device__ __forceinline__ long long lane_prefix_count(const Rec* __restrict__ recs, long long a, long long z, long long bound) {
  long long lo = a, hi = z;  // answer in [lo, hi]
  while (hi - lo > 3) {
    const long long q = (hi - lo) >> 2;
    const long long m1 = lo + q, m2 = lo + 2 * q, m3 = lo + 3 * q;
    const bool l1 = (long long)recs[m1].eid < bound, l2 = (long long)recs[m2].eid < bound, l3 = (long long)recs[m3].eid < bound;
    if (l3) lo = m3 + 1;
    else if (l2) { lo = m2 + 1; hi = m3; }
    else if (l1) { lo = m1 + 1; hi = m2; }
    else hi = m1;
  }
  while (lo < hi && (long long)recs[lo].eid < bound) ++lo;
  return lo - a;
}

// The same with a HINT: where node n's visible prefix ended the last time somebody looked (LookupArgs::cursor).  Batch boundaries
// only move forward within an epoch and a node gains a handful of entries per batch, so the answer is the hint itself or a few
// entries to its right: two probes (one round trip) confirm it, a gallop finds it otherwise -- instead of the ~log4(degree)
// dependent rounds of a search from scratch.  A hint that does not fit (another epoch, never set) costs nothing but the
// two probes.  Returns the ABSOLUTE index of the first record with eid >= bound.
__device__ __forceinline__ long long lane_prefix_end_hinted(const Rec* __restrict__ recs, long long ra, long long rz, long long bound, long long hint) {
  long long lo = ra, hi = rz;  // answer in [lo, hi]
  if (hint >= ra && hint <= rz) {
    const bool below = hint == ra || (long long)recs[hint - 1].eid < bound;  // everything left of the hint is visible
    const bool at = hint < rz && (long long)recs[hint].eid < bound;         // ... and so is the entry at the hint
    if (below && !at) return hint;
    if (below) {  // gallop to the right
      lo = hint + 1;
      long long step = 1;
      while (lo + step <= rz && (long long)recs[lo + step - 1].eid < bound) {
        lo += step;
        step <<= 1;
      }
      hi = lo + step - 1 < rz ? lo + step - 1 : rz;
    } else {
      hi = hint - 1;  // recs[hint - 1] is not visible: the answer is at or left of it
    }
  }
  return lo + lane_prefix_count(recs, lo, hi, bound);
}

template <bool RING, int VEC, int BCAP, bool RIDE>
__global__ __launch_bounds__(256) void lookup_tile_kernel(const LookupArgs a, const UpdateArgs u) {
  extern __shared__ __attribute__((aligned(16))) int lds_eid_all[];
  unsigned bid = blockIdx.x, nblk = gridDim.x;
  if constexpr (RING && RIDE) {
    if (bid < a.side_blocks) {
      update_side_work(u, a.side_stage, (int)bid);
      if (a.tail_blocks) tail_signal(u.barrier, true, 0, 0);
      return;
    }
    bid -= a.side_blocks;
    nblk -= a.side_blocks + a.tail_blocks;
    if (bid >= nblk) {
      tail_commit(u, bid - nblk, a.tail_blocks, nblk + a.side_blocks);
      return;
    }
  }
  const int lane = lane_id();
  const int wave_in_block = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int k = a.k, B = a.B;
  // per wave: ONE staging area of [64, k] ints -- the tile's ids, then the times of rows 0..31, then of rows 32..63 (each flushed to
  // the outputs as flat 16-byte pieces), then the feature-row sources the copy phase looks up -- | the copy phase's list of output
  // slots to write, [64, k] 16-bit entries.  7.9 KB at k = 20 (12 waves per CU fit by registers: 3 per SIMD)
  int* L = lds_eid_all + wave_in_block * (kWave * k + kWave * k / 2 + kWave);
  int* Lstage = L;
  int* Lfirst = L + kWave * k;  // the copy phase's list of output slots to write: [64 * k] 16-bit entries
  const long long tiles = (a.S + kWave - 1) / kWave;
  // normally ONE tile per wave (the launch sizes the grid for it); a launch that must leave room on every CU for the ring
  // update's kernels on the side stream caps the grid, and its waves take a second tile
  for (long long tv = (long long)bid * wpb + wave_in_block; tv < tiles; tv += (long long)nblk * wpb) {
    // the tile index is the same in every lane of the wave: say so (scalar registers; the copy phase's buffer descriptor needs it)
    const long long t = (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)tv >> 32)) << 32) |
                                    __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)tv));
    const long long s = t * kWave + lane;
    const bool act = s < a.S;
    int n = -1, v_old = 0;
    long long q = 0;
    if (act) {
      if (a.grp.groups > 0) {  // hop 0: the seed comes from its group (or is drawn: generated negatives) and is published
        fetch_seed(a, s, 0, true, n, q);
      } else {
        n = a.seeds[s];
        q = a.qtimes[s];
      }
      if (a.out_valid) v_old = a.out_valid[s];  // (delta feature writes) read here: one round trip with the seeds, not one of its own
      int st = 0;
      if (n >= a.N || n < -1 || (n == -1 && !a.allow_pad)) st |= TGMX_ST_SEED_RANGE;
      if (q < 0 && !a.allow_pad) st |= TGMX_ST_SEED_TIME;
      if (st) atomicOr(a.status, st);
    }
    const bool live = n >= 0 && n < a.N;
    // the lane's window: records [w0, w0 + wlen), oldest -> newest after rotating by wrot (rings)
    long long w0 = 0;
    int wrot = 0, wlen = 0;
    if (live) {
      if constexpr (RING) {
        w0 = (long long)n * B;
        wlen = B;
      } else {
        const long long ra = a.indptr[n], rz = a.indptr[n + 1];
        const long long hint = a.cursor ? a.cursor[n] : -1;  // (one round trip with the two loads above)
        const long long p_hi = lane_prefix_end_hinted(a.recs, ra, rz, a.ev_hi, hint);
        if (a.cursor && p_hi != hint) a.cursor[n] = p_hi;  // lanes that share the node store the same value
        const long long p_lo = a.ev_lo <= 0 ? ra : ra + lane_prefix_count(a.recs, ra, rz, a.ev_lo);
        w0 = p_hi - B > p_lo ? p_hi - B : p_lo;
        wlen = (int)(p_hi - w0);
      }
    }
    Rec r[BCAP];
#pragma unroll
    for (int j = 0; j < BCAP; ++j) {
      r[j].nbr = -1;
      r[j].eid = 0;
      r[j].ts = 0;
      if (j < wlen) r[j] = a.recs[w0 + j];
    }
    if constexpr (RING) {
      if (live) wrot = a.write_pos[n] % B;  // independent of the record loads above: one round trip for both
    }
    int cnt = 0;  // 1 + unrolled position of the newest entry with ts < q
#pragma unroll
    for (int j = 0; j < BCAP; ++j) {
      int i = j - wrot;
      if (i < 0) i += B;
      if (j < wlen && r[j].nbr >= 0 && r[j].ts < q && i >= cnt) cnt = i + 1;
      // keep the record ONE 16-byte load: rings never use `eid`, and hipcc would split the load into a dword and a dwordx2 --
      // twice the instructions, each walking 64 scattered lines through the CU's address unit.  (An empty use, placed where the
      // record is consumed anyway: next to the load it would put a wait behind every load.)
      if constexpr (RING) asm volatile("" ::"v"(r[j].eid));
    }
    for (int c = 0; c < k; ++c) {  // pads; the entries below overwrite what the window fills
      Lstage[lane * k + c] = -1;
    }
    // the output slot of window record j (-1: not part of the row -- newer than q, older than the k kept, or a pad record);
    // everything below works from these and the records' payload
    int cj[BCAP];
    int minc = k;  // leftmost non-pad output slot (see lookup_seed: the row's SPAN, an interior pad record stays a pad)
#pragma unroll
    for (int j = 0; j < BCAP; ++j) {
      int i = j - wrot;
      if (i < 0) i += B;
      const int c = i - (cnt - k);  // output slot of unrolled position i
      cj[j] = (j < wlen && i < cnt && c >= 0 && r[j].nbr >= 0) ? c : -1;
      if (cj[j] >= 0) {
        Lstage[lane * k + c] = r[j].nbr;
        minc = c < minc ? c : minc;
      }
    }
    int first_slot = act ? 0 : k;
    if (a.out_valid && act) {  // delta feature writes: see LookupArgs::out_valid
      const int v_new = k - minc;
      first_slot = k - (v_old > v_new ? v_old : v_new);
      a.out_valid[s] = v_new;
      if (a.out_valid_prev) a.out_valid_prev[s] = v_old;
    }
    __builtin_amdgcn_wave_barrier();
    // flush the tile's ids: flat 16-byte pieces of [rows, k]
    const int rows = (int)((a.S - t * kWave) < kWave ? (a.S - t * kWave) : kWave);
    const int ne = rows * k;
    {
      const int4* __restrict__ L4 = reinterpret_cast<const int4*>(Lstage);
      int32_t* __restrict__ G = a.out_nid + t * kWave * k;
      int4* __restrict__ G4 = reinterpret_cast<int4*>(G);
      for (int f = lane; f < (ne >> 2); f += kWave) G4[f] = L4[f];
      for (int f = (ne & ~3) + lane; f < ne; f += kWave) G[f] = Lstage[f];
    }
    // the times, 32 rows at a time through the same staging area ([32, k] int64 = [64, k] ints)
    long long* Lts = reinterpret_cast<long long*>(Lstage);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __builtin_amdgcn_wave_barrier();  // the staging area's previous contents have been read
      const bool mine = (lane >> 5) == half;
      const int lr = lane & 31;
      if (mine) {
        for (int c = 0; c < k; ++c) Lts[lr * k + c] = 0;
#pragma unroll
        for (int j = 0; j < BCAP; ++j)
          if (cj[j] >= 0) Lts[lr * k + cj[j]] = r[j].ts;
      }
      __builtin_amdgcn_wave_barrier();
      const int hrows = rows - 32 * half < 32 ? rows - 32 * half : 32;  // rows of this half that exist
      if (hrows > 0) {
        const int he = hrows * k;
        const int4* __restrict__ T4 = reinterpret_cast<const int4*>(Lts);
        int64_t* __restrict__ H = a.out_ts + (t * kWave + 32 * half) * k;
        int4* __restrict__ H4 = reinterpret_cast<int4*>(H);
        for (int f = lane; f < (he >> 1); f += kWave) H4[f] = T4[f];
        if ((he & 1) && lane == 0) H[he - 1] = Lts[he - 1];
      }
    }
    if (a.D > 0) {
      // the staging area's last tenant: the feature row every output slot is copied from (-1: zeros)
      __builtin_amdgcn_wave_barrier();
      for (int c = 0; c < k; ++c) Lstage[lane * k + c] = -1;
#pragma unroll
      for (int j = 0; j < BCAP; ++j)
        if (cj[j] >= 0) Lstage[lane * k + cj[j]] = (RING || a.x_by_pos) ? (int)(w0 + j) : r[j].eid;
      // ... and the list of output slots that need a write at all: an exclusive scan of the rows' counts over the lanes
      const int nact = k - first_slot;  // first_slot = k for rows beyond S
      int incl = nact;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      const int n_rows = __shfl(incl, kWave - 1);
      unsigned short* Llist = reinterpret_cast<unsigned short*>(Lfirst);
      for (int i = 0; i < nact; ++i) Llist[incl - nact + i] = (unsigned short)((lane << 8) | (first_slot + i));
      __builtin_amdgcn_wave_barrier();
      // (rings: the feature table is < 4 GB -- checked at launch; a static index may cover any number of edges)
      if constexpr (RING) tile_copy<VEC, true>(a, t, rows, k, lane, Lstage, Llist, n_rows, (long long)a.N * B);
      else tile_copy<VEC, false>(a, t, rows, k, lane, Lstage, Llist, n_rows, 0);
    }
    __builtin_amdgcn_wave_barrier();  // the next tile reuses the staging area
  }
  if constexpr (RING && RIDE) {
    if (a.tail_blocks) tail_signal(u.barrier, false, bid, nblk);
  }
}

// ---- the tile kernel with a COOPERATIVE index phase (round 6) -------------------------------------------------------------------
// lookup_tile_kernel's index phase reads a seed's window as BCAP loads of ONE lane: every load instruction of the wave touches 64
// different lines for 16 bytes each (the calibration's "16-byte chunks" pattern, 1.1 TB/s; the lines are re-touched by the next
// seven instructions and mostly hit in L1 / L2, which is what lifts the phase to 2.5-3.5 TB/s) -- against 4.75 TB/s for the same
// windows read as 320 CONTIGUOUS bytes by neighbouring lanes (tools/gather_calib.hip, profiles/r05_narrow_row_calibration.md).
// Here the windows arrive that way: load instruction i serves SPI = 64 / BCAP seeds, BCAP consecutive lanes per seed, lane's record
// j = lane % BCAP -- 3 windows = ~10 lines per instruction at B = 20 instead of 64.  The pick needs no transposition: a record's
// lane decides "valid and older than q" for ITSELF, one ballot per instruction gives every lane the bit field of its seed's window
// (rotated into time order by the ring's write position with two shifts), and the newest qualifying position, the record's own
// output slot and the row's span are bit arithmetic on that field -- computed redundantly by the BCAP lanes of a seed, no shuffles.
// Per-seed scalars (window base, q, write position) go from the seed's lane to its record lanes through a small LDS header.
// Everything after the pick (staging, coalesced flushes, delta-aware copy phase) is lookup_tile_kernel's, fed from the same LDS
// areas; results are identical (tests: every sampler test runs both, TGMX_TILE_COOP=0 is the A/B knob).  BCAP <= 20 (B <= 20).
// Registers: hipcc hoists everything that depends on the lane alone out of the tile loop (first version: 150 registers of LDS addresses,
// one wave per SIMD); `fresh` below makes every phase recompute them.  Streaming rings WITHOUT riders (the comment-shaped step: the
// update's front half is a 1 024-thread workgroup on the library's side stream) are held to TWO waves per SIMD on purpose: with three
// (forced, 145 registers) the same kernel takes 107 instead of 90-99 us per launch on the comment shape -- the front-half workgroup then
// finds no CU with a quarter of its registers free until tiles retire (profiles/r06_tile_coop_occupancy_sweep.txt).  The static
// index has no such neighbour and runs at its natural 155 registers (three waves per SIMD; two measure the same).
template <bool RING, int VEC, int BCAP, bool RIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, (RING && !RIDE) ? 2 : 8))) void lookup_tile_coop_kernel(const LookupArgs a, const UpdateArgs u) {
  extern __shared__ __attribute__((aligned(16))) int lds_eid_all[];
  constexpr int SPI = kWave / BCAP;                  // seeds per load instruction
  constexpr int NI = (kWave + SPI - 1) / SPI;        // load instructions per tile
  unsigned bid = blockIdx.x, nblk = gridDim.x;
  if constexpr (RING && RIDE) {
    if (bid < a.side_blocks) {
      update_side_work(u, a.side_stage, (int)bid);
      if (a.tail_blocks) tail_signal(u.barrier, true, 0, 0);
      return;
    }
    bid -= a.side_blocks;
    nblk -= a.side_blocks + a.tail_blocks;
    if (bid >= nblk) {
      tail_commit(u, bid - nblk, a.tail_blocks, nblk + a.side_blocks);
      return;
    }
  }
  const int lane = lane_id();
  const int wave_in_block = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int k = a.k, B = a.B;
  // per wave: lookup_tile_kernel's staging area + list area, then the header: [64] x {w0 lo, w0 hi, wlen, wrot} | [64] q | [64] span
  constexpr int kHdrInts = kWave * 4 + kWave * 2 + kWave;
  int* L = lds_eid_all + wave_in_block * (kWave * k + kWave * k / 2 + kWave + kHdrInts);
  int* Lstage = L;
  int* Lfirst = L + kWave * k;
  int4* Lhdr = reinterpret_cast<int4*>(L + kWave * k + kWave * k / 2 + kWave);  // (the three areas before it are multiples of 16 bytes: k even or not, 64 k ints are)
  long long* Lq = reinterpret_cast<long long*>(Lhdr + kWave);
  int* Lspan = reinterpret_cast<int*>(Lq + kWave);
  // this lane as a RECORD lane: record j of the seg-th seed of every load instruction
  // (hipcc hoists everything that depends on the lane alone out of the tile loop -- here that is ~150 registers of per-instruction LDS
  // addresses, live across the whole kernel: one wave per SIMD.  `fresh` hands a phase its own copy of the segment index, so the
  // addresses are recomputed where they are used: a handful of integer operations per record)
  const int seg0 = lane / BCAP;
  auto fresh = [](int v) __attribute__((always_inline)) {
    asm volatile("" : "+v"(v));
    return v;
  };
  const unsigned FM = (1u << BCAP) - 1, BM = (1u << B) - 1;  // (BCAP, B <= 20: a window's bit field fits 32 bits)
  const long long tiles = (a.S + kWave - 1) / kWave;
  for (long long tv = (long long)bid * wpb + wave_in_block; tv < tiles; tv += (long long)nblk * wpb) {
    const long long t = (long long)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)tv >> 32)) << 32) |
                                    __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)tv));
    const long long s = t * kWave + lane;
    const bool act = s < a.S;
    int n = -1, v_old = 0;
    long long q = 0;
    if (act) {
      if (a.grp.groups > 0) {
        fetch_seed(a, s, 0, true, n, q);
      } else {
        n = a.seeds[s];
        q = a.qtimes[s];
      }
      if (a.out_valid) v_old = a.out_valid[s];
      int st = 0;
      if (n >= a.N || n < -1 || (n == -1 && !a.allow_pad)) st |= TGMX_ST_SEED_RANGE;
      if (q < 0 && !a.allow_pad) st |= TGMX_ST_SEED_TIME;
      if (st) atomicOr(a.status, st);
    }
    const bool live = n >= 0 && n < a.N;
    long long w0 = 0;
    int wlen = 0;
    if (live) {
      if constexpr (RING) {
        w0 = (long long)n * B;
        wlen = B;
      } else {
        const long long ra = a.indptr[n], rz = a.indptr[n + 1];
        const long long hint = a.cursor ? a.cursor[n] : -1;
        const long long p_hi = lane_prefix_end_hinted(a.recs, ra, rz, a.ev_hi, hint);
        if (a.cursor && p_hi != hint) a.cursor[n] = p_hi;
        const long long p_lo = a.ev_lo <= 0 ? ra : ra + lane_prefix_count(a.recs, ra, rz, a.ev_lo);
        w0 = p_hi - B > p_lo ? p_hi - B : p_lo;
        wlen = (int)(p_hi - w0);
      }
    }
    int wpos = 0;
    if constexpr (RING) {
      if (live) wpos = a.write_pos[n];  // in flight beside the record loads below; consumed after them
    }
    Lhdr[lane] = int4{(int)(unsigned)(unsigned long long)w0, (int)(unsigned)((unsigned long long)w0 >> 32), wlen, 0};
    Lq[lane] = q;
    for (int c = 0; c < k; ++c) Lstage[lane * k + c] = -1;  // pads; the pick below overwrites what the windows fill
    __builtin_amdgcn_wave_barrier();
    // ---- the windows, cooperatively: instruction i = seeds [i SPI, i SPI + SPI), BCAP lanes each
    int seg = fresh(seg0), j = lane - seg * BCAP;
    bool rec_lane = seg < SPI;
    // Branch-free: NI unconditional loads -- a lane without a record (a pad / absent seed, the wave's spare lanes) reads record 0 (a store
    // with an edge has one) and is masked in the pick.
    Rec r[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int sl = i * SPI + seg;
      const int4 h = Lhdr[sl < kWave ? sl : kWave - 1];
      const long long base = (long long)(((unsigned long long)(unsigned)h.y << 32) | (unsigned)h.x);
      r[i] = a.recs[(rec_lane && sl < kWave && j < h.z) ? base + j : 0];
    }
    if constexpr (RING) {
      Lhdr[lane].w = live ? wpos % B : 0;
      __builtin_amdgcn_wave_barrier();
    }
    // ---- the pick: per instruction one ballot, the rest is arithmetic on the seed's bit field
    seg = fresh(seg0), j = lane - seg * BCAP, rec_lane = seg < SPI;
    const int sh = seg * BCAP;
    // output slot of the lane's record of instruction i, + 1 (0: not part of the row), four to a register
    unsigned cs4[(NI + 3) / 4];
#pragma unroll
    for (int w = 0; w < (NI + 3) / 4; ++w) cs4[w] = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int sl = i * SPI + seg, slc = sl < kWave ? sl : kWave - 1;
      const long long qq = Lq[slc];
      const int4 h = Lhdr[slc];
      const int wrot = RING ? h.w : 0;
      const bool has = rec_lane && sl < kWave && j < h.z && r[i].nbr >= 0;
      const unsigned long long m_ok = __ballot(has && r[i].ts < qq), m_nb = __ballot(has);
      if constexpr (RING) asm volatile("" ::"v"(r[i].eid));  // keep the record ONE 16-byte load (see lookup_tile_kernel)
      unsigned f_ok = (unsigned)(m_ok >> sh) & FM, f_nb = (unsigned)(m_nb >> sh) & FM;  // bit jj = record jj of this lane's seed
      if constexpr (RING) {  // unrolled (oldest -> newest) position of slot jj is (jj - wrot) mod B: rotate the fields right by wrot
        f_ok = ((f_ok >> wrot) | (f_ok << (B - wrot))) & BM;
        f_nb = ((f_nb >> wrot) | (f_nb << (B - wrot))) & BM;
      }
      const int cnt = f_ok ? 32 - __clz((int)f_ok) : 0;  // 1 + unrolled position of the newest entry with ts < q
      int iu = j - wrot;
      if (iu < 0) iu += B;
      const int c = iu - (cnt - k);  // output slot of unrolled position iu
      const bool valid = has && iu < cnt && c >= 0;
      cs4[i >> 2] |= (valid ? (unsigned)(c + 1) : 0u) << (8 * (i & 3));
      if (valid) Lstage[sl * k + c] = r[i].nbr;
      if (rec_lane && sl < kWave && j == 0) {
        // the row's SPAN (lookup_seed): k - its leftmost non-pad output slot; the window's positions [cnt - k, cnt) feed slots [0, k)
        const int lo = cnt - k > 0 ? cnt - k : 0;
        const unsigned w = (f_nb & ((1u << cnt) - 1)) >> lo;
        Lspan[sl] = w ? k - (lo + __builtin_ctz(w) - (cnt - k)) : 0;
      }
    }
    __builtin_amdgcn_wave_barrier();
    int first_slot = act ? 0 : k;
    if (a.out_valid && act) {  // delta feature writes: see LookupArgs::out_valid
      const int v_new = Lspan[lane];
      first_slot = k - (v_old > v_new ? v_old : v_new);
      a.out_valid[s] = v_new;
      if (a.out_valid_prev) a.out_valid_prev[s] = v_old;
    }
    // flush the tile's ids: flat 16-byte pieces of [rows, k]
    const int rows = (int)((a.S - t * kWave) < kWave ? (a.S - t * kWave) : kWave);
    const int ne = rows * k;
    {
      const int4* __restrict__ L4 = reinterpret_cast<const int4*>(Lstage);
      int32_t* __restrict__ G = a.out_nid + t * kWave * k;
      int4* __restrict__ G4 = reinterpret_cast<int4*>(G);
      for (int f = lane; f < (ne >> 2); f += kWave) G4[f] = L4[f];
      for (int f = (ne & ~3) + lane; f < ne; f += kWave) G[f] = Lstage[f];
    }
    // the times, 32 rows at a time through the same staging area ([32, k] int64 = [64, k] ints)
    long long* Lts = reinterpret_cast<long long*>(Lstage);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __builtin_amdgcn_wave_barrier();  // the staging area's previous contents have been read
      if ((lane >> 5) == half) {
        const int lr = lane & 31;
        for (int c = 0; c < k; ++c) Lts[lr * k + c] = 0;
      }
      __builtin_amdgcn_wave_barrier();
      seg = fresh(seg0);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int sl = i * SPI + seg;
        const int c1 = (int)((cs4[i >> 2] >> (8 * (i & 3))) & 255u);
        if (c1 > 0 && (sl >> 5) == half) Lts[(sl & 31) * k + c1 - 1] = r[i].ts;
      }
      __builtin_amdgcn_wave_barrier();
      const int hrows = rows - 32 * half < 32 ? rows - 32 * half : 32;
      if (hrows > 0) {
        const int he = hrows * k;
        const int4* __restrict__ T4 = reinterpret_cast<const int4*>(Lts);
        int64_t* __restrict__ H = a.out_ts + (t * kWave + 32 * half) * k;
        int4* __restrict__ H4 = reinterpret_cast<int4*>(H);
        for (int f = lane; f < (he >> 1); f += kWave) H4[f] = T4[f];
        if ((he & 1) && lane == 0) H[he - 1] = Lts[he - 1];
      }
    }
    if (a.D > 0) {
      // the staging area's last tenant: the feature row every output slot is copied from (-1: zeros)
      __builtin_amdgcn_wave_barrier();
      for (int c = 0; c < k; ++c) Lstage[lane * k + c] = -1;
      __builtin_amdgcn_wave_barrier();
      seg = fresh(seg0), j = lane - seg * BCAP;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int sl = i * SPI + seg;
        const int c1 = (int)((cs4[i >> 2] >> (8 * (i & 3))) & 255u);
        if (c1 > 0) Lstage[sl * k + c1 - 1] = (RING || a.x_by_pos) ? (int)((unsigned)Lhdr[sl].x + (unsigned)j) : r[i].eid;
      }
      const int nact = k - first_slot;
      int incl = nact;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      const int n_rows = __shfl(incl, kWave - 1);
      unsigned short* Llist = reinterpret_cast<unsigned short*>(Lfirst);
      for (int i = 0; i < nact; ++i) Llist[incl - nact + i] = (unsigned short)((lane << 8) | (first_slot + i));
      __builtin_amdgcn_wave_barrier();
      // (the copy loop's per-lane constants -- ~40 registers of piece columns and offsets -- are recomputed per tile: hoisted out of the tile
      // loop they would be live across the index phase, whose NI records already fill a third of the register file)
      int lane_c = lane;
      asm volatile("" : "+v"(lane_c));
      if constexpr (RING) tile_copy<VEC, true>(a, t, rows, k, lane_c, Lstage, Llist, n_rows, (long long)a.N * B);
      else tile_copy<VEC, false>(a, t, rows, k, lane_c, Lstage, Llist, n_rows, 0);
    }
    __builtin_amdgcn_wave_barrier();  // the next tile reuses the staging area and the header
  }
  if constexpr (RING && RIDE) {
    if (a.tail_blocks) tail_signal(u.barrier, false, bid, nblk);
  }
}

// > 64 KB of LDS per workgroup (static + dynamic) is a per-DEVICE opt-in of the kernel function
template <auto Kernel>
static bool lds_optin() {
  constexpr int kMaxDevices = 64;
  static std::atomic<bool> done[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
  if (dev >= 0 && done[dev].load(std::memory_order_acquire)) return true;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  if (e != hipSuccess) {
    set_error("recency_lookup: cannot reserve LDS: %s", hipGetErrorString(e));
    return false;
  }
  if (dev >= 0) done[dev].store(true, std::memory_order_release);
  return true;
}

static int device_cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// A launch that may be timed: with an event pair the kernel goes through hipExtLaunchKernelGGL, whose events carry the
// dispatch's OWN begin / end timestamps (what rocprofv3 --kernel-trace reports); hipEventRecord before / after the launch
// brackets it from outside and adds the command processor's event handling (~4-6 us under a full queue)
#define TGMX_LAUNCH_TIMED(KERNEL, GRID, BLOCK, LDS, STREAM, E0, E1, ...)                          \
  do {                                                                                           \
    if ((E0) && (E1)) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, (uint32_t)(LDS), STREAM, E0, E1, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                      \
  } while (0)

static int prepare_lookup(LookupArgs& a, const float* out_x, int kmax) {
  int vec = 1;
  if (a.D > 0) {
    const uintptr_t bits = (uintptr_t)a.edge_x | (uintptr_t)out_x | (uintptr_t)a.out_x;
    if (a.D % 4 == 0 && (bits & 15) == 0) vec = 4;
    else if (a.D % 2 == 0 && (bits & 7) == 0) vec = 2;
  }
  a.row_vecs = a.D > 0 ? a.D / vec : 1;
  a.dv = make_fastdiv((uint32_t)a.row_vecs);
  a.dv_seed = make_fastdiv((uint32_t)(kmax * a.row_vecs));
  if ((unsigned long long)kmax * a.row_vecs * (unsigned long long)a.row_vecs >= (1ull << 32)) {
    set_error("recency_lookup: k*D^2 too large for the slot divider (k=%d, D=%d)", kmax, a.D);
    return TGMX_E_UNSUPPORTED;
  }
  return vec;
}

// lanes per seed of the packed kernel for (B, k); 64 = the packed kernel does not apply
static int packed_group_lanes(const LookupArgs& a, int k, bool ring) {
  const int gl = (a.B <= 16 && k <= 16) ? 16 : ((a.B <= 32 && k <= 32) ? 32 : 64);
  return (ring && gl < 64 && (long long)k * a.row_vecs <= 8 * gl) ? gl : 64;
}

template <bool RING>
static int launch_lookup(LookupArgs a, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop,
                         const UpdateArgs* side = nullptr, int side_stage = 0, unsigned side_blocks = 0, unsigned tail_blocks = 0) {
  if (a.S == 0) return TGMX_OK;
  const UpdateArgs u = side ? *side : UpdateArgs{};
  a.side_blocks = (RING && side) ? side_blocks : 0;
  a.side_stage = side_stage;
  a.tail_blocks = a.side_blocks ? tail_blocks : 0;
  const bool small = a.B <= kWave && a.k <= kWave;
  const int vec = prepare_lookup(a, a.out_x, a.k);
  if (vec < 0) return vec;
  const int waves_per_block = 4;
  long long blocks = (a.S + waves_per_block - 1) / waves_per_block;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  const dim3 grid((unsigned)blocks + a.side_blocks + a.tail_blocks), block(waves_per_block * kWave);
  const size_t lds = (size_t)waves_per_block * a.k * sizeof(int);
  const bool ride = RING && a.side_blocks > 0;
#define TGMX_LAUNCH(VEC_, SMALL_)                                                                                              \
  do {                                                                                                                         \
    if (ride) TGMX_LAUNCH_TIMED((recency_lookup_kernel<RING, VEC_, SMALL_, true>), grid, block, lds, stream, ev_start, ev_stop, a, u); \
    else TGMX_LAUNCH_TIMED((recency_lookup_kernel<RING, VEC_, SMALL_, false>), grid, block, lds, stream, ev_start, ev_stop, a, u);     \
  } while (0)
  // narrow rows: several seeds per wave (streaming rings, plain seed arrays)
  const int gl_any = a.out_eid ? 64 : packed_group_lanes(a, a.k, true);  // narrow rows?  (edge ids are written by the wave-per-seed kernels only)
  static const bool packed_hop0 = !(getenv("TGMX_PACKED_HOP0") && atoi(getenv("TGMX_PACKED_HOP0")) == 0);  // A/B knob
  const int gl = (a.grp.groups == 0 || packed_hop0) ? gl_any : 64;  // (round 3: the packed kernel takes seed groups -- hop 0 -- too)
  static const bool tile_on = !(getenv("TGMX_TILE") && atoi(getenv("TGMX_TILE")) == 0);  // A/B knob: 0 = the packed kernel
  // A tile's phases are a chain of ~8 dependent round trips (15-20 us on an idle chip): it pays when the launch has at least two tiles
  // per CU to overlap them (comment shape, hop 1: 3840 tiles), not for a few hundred seeds -- measured: review shape, hop 0 (24
  // tiles) 6 -> 19 us, hop 1 (240 tiles) 17.7 -> 19.5 us; comment shape, hop 0 (192 tiles) slower too.  Small launches keep the
  // kernels that give every seed (group) a wave of its own.
  const bool tile_ok = gl_any < 64 && tile_on && a.B <= 32 && a.k <= 32 && (((uintptr_t)a.out_nid | (uintptr_t)a.out_ts) & 15) == 0 &&
                       a.S <= (1ll << 25) && a.S >= 2ll * kWave * device_cu_count() &&
                       (!RING || (long long)a.N * a.B * (a.D > 0 ? a.D : 1) * 4 < (1ll << 32) - 64);
  if (tile_ok) {
    // a tile of 64 seeds per wave; without riders one wave per workgroup (finest balance, every tile resident at once)
    const int wpb = ride ? 4 : 1;
    const long long tiles = (a.S + kWave - 1) / kWave;
    long long tblocks = (tiles + wpb - 1) / wpb;  // one tile per wave (tile_ok: fits a grid)
    if (a.leave_room && !ride) {
      // Every tile resident at once means 15-16 waves of 128 registers per CU: nothing else can be dispatched until tiles
      // retire, and the side stream's dozen small kernels (the large ring update's front half) then run AFTER the lookups instead
      // of beside them (measured, comment shape: lookup 128 -> 112 us but the step 179 -> 202 us).  12 waves per CU leave a
      // quarter of every SIMD's registers and half the LDS free; the surplus tiles are second tiles of the first waves.
      const long long room = (long long)device_cu_count() * 12;
      if (tblocks > room) tblocks = room;
    }
    const dim3 tgrid((unsigned)tblocks + a.side_blocks + a.tail_blocks), tblock(wpb * kWave);
    const int bcap = a.B <= 10 ? 10 : (a.B <= 20 ? 20 : 32);
    // the cooperative index phase (windows read as contiguous runs by neighbouring lanes, lookup_tile_coop_kernel): B <= 20, the low 32 bits
    // of a window's base must address it (rings: checked above; static index: fewer than 2^31 records)
    static const bool coop_on = !(getenv("TGMX_TILE_COOP") && atoi(getenv("TGMX_TILE_COOP")) == 0);  // A/B knob: 0 = one lane per window
    const bool coop = coop_on && bcap <= 20;
    const size_t tlds = (size_t)wpb * (kWave * a.k + kWave * a.k / 2 + kWave + (coop ? kWave * 7 : 0)) * sizeof(int);
#define TGMX_TILE_LAUNCH_K(KERNEL_, VEC_, BCAP_)                                                                                            \
  do {                                                                                                                                      \
    if (ride) {                                                                                                                             \
      if (tlds > 32 * 1024 && !lds_optin<KERNEL_<RING, VEC_, BCAP_, RING>>()) return TGMX_E_LAUNCH;                                          \
      TGMX_LAUNCH_TIMED((KERNEL_<RING, VEC_, BCAP_, RING>), tgrid, tblock, tlds, stream, ev_start, ev_stop, a, u);                           \
    } else {                                                                                                                                \
      if (tlds > 32 * 1024 && !lds_optin<KERNEL_<RING, VEC_, BCAP_, false>>()) return TGMX_E_LAUNCH;                                         \
      TGMX_LAUNCH_TIMED((KERNEL_<RING, VEC_, BCAP_, false>), tgrid, tblock, tlds, stream, ev_start, ev_stop, a, u);                          \
    }                                                                                                                                       \
  } while (0)
#define TGMX_TILE_LAUNCH(VEC_, BCAP_) TGMX_TILE_LAUNCH_K(lookup_tile_kernel, VEC_, BCAP_)
#define TGMX_TILE_VEC(VEC_)                                                           \
  do {                                                                                \
    if (bcap == 10 && coop) TGMX_TILE_LAUNCH_K(lookup_tile_coop_kernel, VEC_, 10);    \
    else if (bcap == 20 && coop) TGMX_TILE_LAUNCH_K(lookup_tile_coop_kernel, VEC_, 20); \
    else if (bcap == 10) TGMX_TILE_LAUNCH(VEC_, 10);                                  \
    else if (bcap == 20) TGMX_TILE_LAUNCH(VEC_, 20);                                  \
    else TGMX_TILE_LAUNCH(VEC_, 32);                                                  \
  } while (0)
    if (vec == 4) TGMX_TILE_VEC(4);
    else {
      if (vec == 2) {  // no 8-byte instantiation: float pieces
        a.row_vecs = a.D;
        a.dv = make_fastdiv((uint32_t)a.row_vecs);
        a.dv_seed = make_fastdiv((uint32_t)(a.k * a.row_vecs));
      }
      TGMX_TILE_VEC(1);
    }
#undef TGMX_TILE_VEC
#undef TGMX_TILE_LAUNCH
#undef TGMX_TILE_LAUNCH_K
  } else if (gl < 64) {
    const int per_wave = 64 / gl;
    long long pblocks = ((a.S + per_wave - 1) / per_wave + waves_per_block - 1) / waves_per_block;
    if (pblocks > (1 << 20)) pblocks = 1 << 20;
    const dim3 pgrid((unsigned)pblocks + a.side_blocks + a.tail_blocks);
    const size_t plds = (size_t)waves_per_block * per_wave * a.k * sizeof(int);
#define TGMX_PACKED(VEC_)                                                                              \
  do {                                                                                                 \
    if (gl == 16 && ride) TGMX_LAUNCH_TIMED((lookup_packed_kernel<RING, VEC_, 16, true>), pgrid, block, plds, stream, ev_start, ev_stop, a, u); \
    else if (gl == 16) TGMX_LAUNCH_TIMED((lookup_packed_kernel<RING, VEC_, 16, false>), pgrid, block, plds, stream, ev_start, ev_stop, a, u);  \
    else if (ride) TGMX_LAUNCH_TIMED((lookup_packed_kernel<RING, VEC_, 32, true>), pgrid, block, plds, stream, ev_start, ev_stop, a, u);       \
    else TGMX_LAUNCH_TIMED((lookup_packed_kernel<RING, VEC_, 32, false>), pgrid, block, plds, stream, ev_start, ev_stop, a, u);                \
  } while (0)
    if (vec == 4) TGMX_PACKED(4);
    else if (vec == 2) TGMX_PACKED(2);
    else TGMX_PACKED(1);
#undef TGMX_PACKED
  } else if (small) {
    if (vec == 4) TGMX_LAUNCH(4, true);
    else if (vec == 2) TGMX_LAUNCH(2, true);
    else TGMX_LAUNCH(1, true);
  } else {
    if (vec == 4) TGMX_LAUNCH(4, false);
    else if (vec == 2) TGMX_LAUNCH(2, false);
    else TGMX_LAUNCH(1, false);
  }
#undef TGMX_LAUNCH
  TGMX_CHECK_LAUNCH("recency_lookup");
  return TGMX_OK;
}


// hop 0 (a: seeds / groups, k, outputs) and hop 1 (k1, out_*1) as one launch; the caller checked can_fuse01
template <bool RING>
static int launch_fused01(LookupArgs a, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, const UpdateArgs* side,
                          int side_stage, unsigned side_blocks, unsigned tail_blocks = 0) {
  const UpdateArgs u = side ? *side : UpdateArgs{};
  a.side_blocks = (RING && side) ? side_blocks : 0;
  a.side_stage = side_stage;
  a.tail_blocks = a.side_blocks ? tail_blocks : 0;
  const int kmax = a.k > a.k1 ? a.k : a.k1;
  const int vec = prepare_lookup(a, a.out_x1, kmax);
  if (vec < 0) return vec;
  const int waves_per_block = 4;
  const long long waves = a.S + a.S * a.k;
  long long blocks = (waves + waves_per_block - 1) / waves_per_block;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  const dim3 grid((unsigned)blocks + a.side_blocks + a.tail_blocks), block(waves_per_block * kWave);
  const size_t lds = (size_t)waves_per_block * kmax * sizeof(int);
  // the riders' static LDS sized for what rides (RiderLds): placement of <= 512 entries, of <= 1024, or sort / merge only
  const int pcap = !(RING && side) ? 1024 : ((side_stage != kSideAll && side_stage != kSideSortMergePlace) ? 0 : (u.m <= 512 ? 512 : 1024));
  if (vec == 4 && pcap == 512) TGMX_LAUNCH_TIMED((recency_lookup_fused01_kernel<RING, 4, RING ? 512 : 1024>), grid, block, lds, stream, ev_start, ev_stop, a, u);
  else if (vec == 4 && pcap == 0) TGMX_LAUNCH_TIMED((recency_lookup_fused01_kernel<RING, 4, RING ? 0 : 1024>), grid, block, lds, stream, ev_start, ev_stop, a, u);
  else if (vec == 4) TGMX_LAUNCH_TIMED((recency_lookup_fused01_kernel<RING, 4>), grid, block, lds, stream, ev_start, ev_stop, a, u);
  else if (vec == 2) TGMX_LAUNCH_TIMED((recency_lookup_fused01_kernel<RING, 2>), grid, block, lds, stream, ev_start, ev_stop, a, u);
  else TGMX_LAUNCH_TIMED((recency_lookup_fused01_kernel<RING, 1>), grid, block, lds, stream, ev_start, ev_stop, a, u);
  TGMX_CHECK_LAUNCH("recency_lookup_fused01");
  return TGMX_OK;
}


// ---- large batches (m > kBlockMaxM, e.g. the replicated update of an 8-rank global batch of 8 x 4096 edges): O(m)
// passes around one rocPRIM radix sort of the (sign-flipped) 64-bit keys -- LSD radix sort is stable, which is the
// reference's `argsort(stable=True)`.  Slot collisions are resolved through an open-addressing hash in the scratch
// (atomicMax of the sorted position: the last one wins); write_pos moves by one atomicAdd per run.
// span = max(ts) + 1, by gridDim.x workgroups meeting in one atomicMax (the word is preset to a very negative value);
// a single workgroup walking 32 768 timestamps took 56 us
__global__ __launch_bounds__(256) void ring_update_span_kernel(const UpdateArgs a) {
  __shared__ long long red[256 / kWave];
  const long long per = (a.n + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < a.n ? lo + per : a.n;
  long long mx = -0x7fffffffffffffffLL;
  if (lo < hi) mx = strided_max_ts(a.ts + lo, hi - lo, threadIdx.x, 256);
  for (int off = 32; off > 0; off >>= 1) {
    const long long o = __shfl_xor(mx, off);
    mx = o > mx ? o : mx;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 256 / kWave; ++w) mx = red[w] > mx ? red[w] : mx;
    if (lo < hi) atomicMax(a.span, mx + 1);
  }
}

__global__ __launch_bounds__(256) void ring_update_keys_kernel(const UpdateArgs a) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.m) return;
  int node, nbr;
  long long t, i;
  update_entry(a, j, node, nbr, t, i);
  // a time-sorted batch (every batch of a chronological loader): max(ts) is its last timestamp -- no reduction launch
  const long long span = a.sorted_ts ? a.ts[a.n - 1] + 1 : *a.span;
  const long long key = update_key(node, t, span, a.key_wrap32);
  if (a.sorted_ts && i + 1 < a.n && a.ts[i + 1] < t) atomicOr(a.status, TGMX_ST_TS_BOUND);  // the promise is checked, not trusted
  if (a.sort_bits < 64) {
    // bounded sort: t in [0, ts_bound] makes key + 2^31 (int32-wrapped product) resp. key (int64 product, valid node)
    // non-negative and narrower than sort_bits
    if (t < 0 || t > a.ts_bound) atomicOr(a.status, TGMX_ST_TS_BOUND);
    a.keys_in[j] = (unsigned long long)(a.key_wrap32 ? key + 2147483648LL : key);
  } else {
    a.keys_in[j] = (unsigned long long)key ^ 0x8000000000000000ull;  // signed -> radix order
  }
  a.vals_in[j] = (unsigned)j;
}

__global__ __launch_bounds__(256) void ring_update_scatter_kernel(const UpdateArgs a) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.m) return;
  int node, nbr;
  long long t, i;
  update_entry(a, a.sorted_j[p], node, nbr, t, i);
  const bool valid = node >= 0 && node < a.N && nbr >= 0 && nbr < a.N;
  if (!valid) atomicOr(a.status, TGMX_ST_EDGE_RANGE);
  a.sorted_node[p] = valid ? node : -1;
}

__device__ __forceinline__ int global_hash_slot(int* keys, int hash_bits, int tgt, bool insert) {
  unsigned h = ((unsigned)tgt * 2654435761u) >> (32 - hash_bits);
  const unsigned mask = (1u << hash_bits) - 1u;
  for (;;) {
    const int old = insert ? atomicCAS(&keys[h], -1, tgt) : keys[h];
    if (old == tgt || (insert && old == -1)) return (int)h;
    h = (h + 1) & mask;
  }
}

// run analysis of the sorted order: run_start by an inclusive max-scan (rocPRIM) of "p if p opens a run else 0",
// run length scattered to the run's first position by the run's last one -- O(m) however long the hub runs are
struct RunFlag {
  const int32_t* sorted_node;
  __device__ __forceinline__ int operator()(int p) const { return (p > 0 && sorted_node[p - 1] == sorted_node[p]) ? 0 : p; }
};

__global__ __launch_bounds__(256) void ring_update_ends_kernel(const UpdateArgs a) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.m) return;
  if (p == a.m - 1 || a.sorted_node[p + 1] != a.sorted_node[p]) a.run_len[a.run_start[p]] = (int)p - a.run_start[p] + 1;
}

// ---- 4096 < m <= 8192 (a comment-shaped batch: 2 x 4096 entries): the state-independent half of the update as ONE workgroup ----
// The generic path above is a chain of 13 small launches (keys, six radix-sort kernels, scatter, scan x 2, ends, ...).  On the side
// stream, beside a lookup launch that saturates the memory system, every one of them crawls (each dependent access costs
// microseconds: 143 us for the chain beside the packed lookup kernel, not finished when the faster tile kernel ends).  Here ONE
// workgroup of 1024 threads sorts the batch in registers / LDS (rocprim::block_radix_sort over the key bits that can be set,
// stable like argsort(stable=True)), and derives the runs with a block scan: compute on one CU, a few coalesced global accesses,
// nothing for the lookups to slow down.  Writes what keys + sort + scatter + scan + ends wrote; the placement launch follows.
constexpr int kFrontIPT = 8, kFrontMaxM = kBlockThreads * kFrontIPT;
__global__ __launch_bounds__(kBlockThreads) void ring_update_front_block_kernel(const UpdateArgs a) {
  using Sort = rocprim::block_radix_sort<unsigned long long, kBlockThreads, kFrontIPT, unsigned>;
  using Scan = rocprim::block_scan<int, kBlockThreads>;
  __shared__ union {
    typename Sort::storage_type sort;
    struct {
      int node[kFrontMaxM + 1];
    } r;
  } L;
  __shared__ typename Scan::storage_type scan_storage;
  __shared__ long long red[kBlockThreads / kWave];
  const int tid = threadIdx.x, m = (int)a.m;
  // the placement's hash: keys and max positions (adjacent), all -1 (was a memset launch)
  for (int i = tid; i < (2 << a.hash_bits); i += kBlockThreads) a.hash_key[i] = -1;
  long long span;
  if (a.sorted_ts) {
    span = a.ts[a.n - 1] + 1;  // a time-sorted batch: max(ts) is its last timestamp
  } else {
    long long mx = strided_max_ts(a.ts, a.n, tid, kBlockThreads);
    for (int off = 32; off > 0; off >>= 1) {
      const long long o = __shfl_xor(mx, off);
      mx = o > mx ? o : mx;
    }
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < kBlockThreads / kWave; ++w) mx = red[w] > mx ? red[w] : mx;
    span = mx + 1;
  }
  unsigned long long key[kFrontIPT];
  unsigned val[kFrontIPT];
#pragma unroll
  for (int e = 0; e < kFrontIPT; ++e) {
    const int j = tid * kFrontIPT + e;
    val[e] = (unsigned)j;
    key[e] = ~0ull;  // padding: sorts behind every entry (stable: equal keys keep entry order, pads have the largest indices)
    if (j < m) {
      int node, nbr;
      long long t, i;
      update_entry(a, j, node, nbr, t, i);
      const long long k = update_key(node, t, span, a.key_wrap32);
      if (a.sorted_ts && i + 1 < a.n && a.ts[i + 1] < t) atomicOr(a.status, TGMX_ST_TS_BOUND);  // the promise is checked, not trusted
      if (a.sort_bits < 64) {
        if (t < 0 || t > a.ts_bound) atomicOr(a.status, TGMX_ST_TS_BOUND);
        key[e] = (unsigned long long)(a.key_wrap32 ? k + 2147483648LL : k);
      } else {
        key[e] = (unsigned long long)k ^ 0x8000000000000000ull;  // signed -> radix order
      }
    }
  }
  Sort().sort(key, val, L.sort, 0u, (unsigned)a.sort_bits);
  __syncthreads();
  // thread tid now holds sorted positions tid * IPT + e
  int node_at[kFrontIPT];
#pragma unroll
  for (int e = 0; e < kFrontIPT; ++e) {
    const int p = tid * kFrontIPT + e;
    node_at[e] = -1;
    if (p < m) {
      int node, nbr;
      long long t, i;
      update_entry(a, val[e], node, nbr, t, i);
      const bool valid = node >= 0 && node < a.N && nbr >= 0 && nbr < a.N;
      if (!valid) atomicOr(a.status, TGMX_ST_EDGE_RANGE);
      node_at[e] = valid ? node : -1;
      a.sorted_j[p] = (int)val[e];
      a.sorted_node[p] = node_at[e];
      L.r.node[p] = node_at[e];
    }
  }
  if (tid == 0) L.r.node[m] = -2;  // sentinel behind the last entry: never equal to a node
  __syncthreads();
  // run_start = inclusive max-scan of "p if p opens a run else 0"; the run's last position scatters the run length to its first
  int flag[kFrontIPT], rs[kFrontIPT];
#pragma unroll
  for (int e = 0; e < kFrontIPT; ++e) {
    const int p = tid * kFrontIPT + e;
    flag[e] = (p < m && p > 0 && L.r.node[p - 1] == node_at[e]) ? 0 : (p < m ? p : 0);
  }
  Scan().inclusive_scan(flag, rs, scan_storage, rocprim::maximum<int>());
#pragma unroll
  for (int e = 0; e < kFrontIPT; ++e) {
    const int p = tid * kFrontIPT + e;
    if (p < m) {
      a.run_start[p] = rs[e];
      if (L.r.node[p + 1] != node_at[e]) a.run_len[rs[e]] = p - rs[e] + 1;
    }
  }
}

__global__ __launch_bounds__(256) void ring_update_place_kernel(const UpdateArgs a) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.m) return;
  const int node = a.sorted_node[p];
  int tgt = -1, kept_at_end = 0;
  if (node >= 0) {
    const int lo = a.run_start[p], cnt = a.run_len[lo], pos = (int)p - lo;
    const int drop = cnt > a.B ? cnt - a.B : 0;
    if (pos >= drop) {
      tgt = node * a.B + (a.write_pos[node] % a.B + pos - drop) % a.B;
      atomicMax(&a.hash_maxp[global_hash_slot(a.hash_key, a.hash_bits, tgt, true)], (int)p);
    }
    if (pos == cnt - 1) kept_at_end = cnt - drop;
  }
  a.target[p] = tgt;
  a.winner[p] = kept_at_end;  // #kept of the run, parked at the run's last position until the write pass
}

__global__ __launch_bounds__(256) void ring_update_write_kernel(const UpdateArgs a) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.m) return;
  if (update_blocked(a)) {
    a.winner[p] = -1;  // nothing placed: the feature copy behind this launch has nothing to do
    return;
  }
  const int tgt = a.target[p];
  const int kept = a.winner[p];
  int win = -1;
  if (tgt >= 0 && a.hash_maxp[global_hash_slot(a.hash_key, a.hash_bits, tgt, false)] == (int)p) {
    int nd, nbr;
    long long t, i;
    update_entry(a, a.sorted_j[p], nd, nbr, t, i);
    Rec r;
    r.nbr = nbr;
    r.eid = a.eid0 >= 0 ? (int)(a.eid0 + i) : -1;
    r.ts = t;
    a.ring[tgt] = r;
    win = tgt;
  }
  if (kept > 0) {  // every write_pos read happened in the placement launch
    int32_t* wp = &a.write_pos[a.sorted_node[p]];
    const int old = atomicAdd(wp, kept);
    constexpr int kFold = 1 << 30;
    if (old < kFold && old + kept >= kFold) atomicSub(wp, kFold / a.B * a.B);
  }
  a.winner[p] = win;
}

// The write pass and the feature rows as ONE launch (D > 0), one wave per sorted position: the position's scalars are wave-uniform
// (scalar loads), lane 0 writes the record and advances write_pos, the wave copies the row.
__global__ __launch_bounds__(256) void ring_update_write_feat_kernel(const UpdateArgs a) {
  const long long p = (long long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (p >= a.m) return;
  const bool first = lane_id() == 0;
  if (update_blocked(a)) {
    if (first) a.winner[p] = -1;
    return;
  }
  const int tgt = a.target[p];
  const int kept = a.winner[p];
  int win = -1;
  long long i = 0;
  if (tgt >= 0 && a.hash_maxp[global_hash_slot(a.hash_key, a.hash_bits, tgt, false)] == (int)p) {
    int nd, nbr;
    long long t;
    update_entry(a, a.sorted_j[p], nd, nbr, t, i);
    if (first) {
      Rec r;
      r.nbr = nbr;
      r.eid = a.eid0 >= 0 ? (int)(a.eid0 + i) : -1;
      r.ts = t;
      a.ring[tgt] = r;
    }
    win = tgt;
  }
  if (first) {
    if (kept > 0) commit_write_pos(&a.write_pos[a.sorted_node[p]], kept, a.B);
    a.winner[p] = win;
  }
  if (win < 0) return;
  float* __restrict__ o = a.ring_x + (long long)win * a.D;
  if (a.edge_x) {
    const float* __restrict__ x = a.edge_x + i * a.D;
    for (int c = lane_id(); c < a.D; c += kWave) o[c] = x[c];
  } else {
    for (int c = lane_id(); c < a.D; c += kWave) o[c] = 0.f;
  }
}

// one wave per sorted position: copy the winning entry's D-float feature row.  COMMIT (after a riding placement, which
// only decided): lane 0 also writes the winner's record and the run's write_pos increment
template <bool COMMIT>
__global__ __launch_bounds__(256) void ring_update_feat_kernel(const UpdateArgs a) {
  const long long p = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= a.m) return;
  if (COMMIT && update_blocked(a)) return;
  const int row = a.winner[p];
  if constexpr (COMMIT) {
    if (lane_id() == 0) {
      const int kept = a.target[p];
      if (row >= 0) a.ring[row] = a.sorted_rec[p];
      if (kept > 0) commit_write_pos(&a.write_pos[a.sorted_node[p]], kept, a.B);
    }
  }
  if (row < 0 || a.D == 0) return;
  const long long j = a.sorted_j[p];
  const long long i = j >= a.n ? j - a.n : j;
  float* __restrict__ o = a.ring_x + (long long)row * a.D;
  if (a.edge_x) {
    const float* __restrict__ x = a.edge_x + i * a.D;
    for (int c = lane_id(); c < a.D; c += kWave) o[c] = x[c];
  } else {
    for (int c = lane_id(); c < a.D; c += kWave) o[c] = 0.f;
  }
}

// 1024 < m <= kBlockMaxM stand-alone (the replicated update of a 4- or 8-rank global batch through `tgmx_ring_update`):
// one workgroup is VALU-bound on the sorting network (37 us at m = 3200), so the sort is spread over the chip with the
// pieces above (chunk sort, then merge with all chunk-sorted keys staged in LDS) and the single-workgroup kernel runs
// with PRESORTED = true (runs, placement, collisions, writes).
__global__ __launch_bounds__(kChunk) void ring_update_chunk_sort_kernel(const UpdateArgs a) {
  __shared__ ChunkSortLds W;
  update_chunk_sort(a, blockIdx.x, W);
}

__global__ __launch_bounds__(kChunk) void ring_update_merge_kernel(const UpdateArgs a) {
  __shared__ long long k_all[kBlockMaxM];
  __shared__ int p_all[kBlockMaxM];
  const int m = (int)a.m;
  const int tid = threadIdx.x;
  for (int base = tid; base < m; base += 8 * kChunk) {  // 16 independent loads in flight per round
    long long kv[8];
    int pv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int x = base + u * kChunk;
      kv[u] = x < m ? a.key[x] : 0;
      pv[u] = x < m ? a.node[x] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int x = base + u * kChunk;
      if (x < m) {
        k_all[x] = kv[u];
        p_all[x] = pv[u];
      }
    }
  }
  __syncthreads();
  update_merge_entry<16>(a, k_all, p_all, blockIdx.x, tid);
}

// scratch of the spread sort: chunk-sorted keys (16-byte aligned int64), sorted records and chunk-sorted entry indices
// live behind the four int32[m] arrays
static unsigned set_chunk_scratch(UpdateArgs& a, int32_t* scratch) {
  const unsigned chunks = (unsigned)((a.m + kChunk - 1) / kChunk);
  long long* s64 = reinterpret_cast<long long*>(((uintptr_t)(scratch + 4 * a.m) + 15) & ~(uintptr_t)15);
  a.key = s64;
  a.sorted_rec = reinterpret_cast<Rec*>(s64 + (long long)chunks * kChunk);
  a.node = reinterpret_cast<int32_t*>(a.sorted_rec + a.m);
  return chunks;
}

// placement of a batch whose sorted order is already in the scratch, then the feature rows
static void launch_update_presorted(const UpdateArgs& a, hipStream_t st) {
  int P = 64;
  while (P < a.m) P <<= 1;
  constexpr unsigned parts = 8;  // measured at m = 3200: 16.2 us (1 workgroup), 11.4 (4), 10.8 (8), 10.7 (16)
  if (P <= 1024) hipLaunchKernelGGL((ring_update_block_kernel<1, 1024, true>), dim3(1), dim3(P), 0, st, a);
  else if (P == 2048) hipLaunchKernelGGL((ring_update_block_kernel<2, 2048, true>), dim3(parts), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL((ring_update_block_kernel<4, 4096, true>), dim3(parts), dim3(1024), 0, st, a);
  if (a.D > 0) hipLaunchKernelGGL(ring_update_feat_kernel<false>, dim3((unsigned)((a.m + 3) / 4)), dim3(256), 0, st, a);
}

// 1024 < m <= 4096 next to the lookups (side stream): chunk sort, merge (+ pre-gather), placement decisions; nothing here
// writes ring state.  The commit (ring_update_feat_kernel<true>) follows the lookups on the main stream.
static void launch_update_mid_front(const UpdateArgs& a, unsigned chunks, hipStream_t st) {
  hipLaunchKernelGGL(ring_update_chunk_sort_kernel, dim3(chunks), dim3(kChunk), 0, st, a);
  hipLaunchKernelGGL(ring_update_merge_kernel, dim3(chunks), dim3(kChunk), 0, st, a);
  int P = 64;
  while (P < a.m) P <<= 1;
  constexpr unsigned parts = 8;
  if (P <= 2048) hipLaunchKernelGGL((ring_update_decide_kernel<2, 2048>), dim3(parts), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL((ring_update_decide_kernel<4, 4096>), dim3(parts), dim3(1024), 0, st, a);
}

static void launch_update_block(UpdateArgs& a, int32_t* scratch, hipStream_t st) {
  int P = 64;
  while (P < a.m) P <<= 1;
  if (P <= 1024) {
    hipLaunchKernelGGL((ring_update_block_kernel<1, 1024, false>), dim3(1), dim3(P), 0, st, a);
    if (a.D > 0) hipLaunchKernelGGL(ring_update_feat_kernel<false>, dim3((unsigned)((a.m + 3) / 4)), dim3(256), 0, st, a);
    return;
  }
  const unsigned chunks = set_chunk_scratch(a, scratch);
  hipLaunchKernelGGL(ring_update_chunk_sort_kernel, dim3(chunks), dim3(kChunk), 0, st, a);
  hipLaunchKernelGGL(ring_update_merge_kernel, dim3(chunks), dim3(kChunk), 0, st, a);
  launch_update_presorted(a, st);
}

__global__ __launch_bounds__(256) void ring_reset_kernel(Rec* ring, int32_t* write_pos, long long nrec, int N) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  Rec pad;
  pad.nbr = -1; pad.eid = 0; pad.ts = 0;
  for (long long x = i; x < nrec; x += step) ring[x] = pad;
  for (long long x = i; x < N; x += step) write_pos[x] = 0;
}

}  // namespace tgmx

using namespace tgmx;

namespace tgmx {

struct UniformArgs {
  const int64_t* indptr;
  const Rec* recs;
  const float* edge_x;
  const int32_t* seeds;
  int32_t* out_nid;
  int64_t* out_ts;
  float* out_x;
  int32_t* status;
  long long S, ev_hi;
  unsigned long long rng_seed, rng_stream;
  int D, k, N, allow_pad;
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// One wave per seed.  c = #candidates (prefix of the node's (eid, role)-ordered entries below ev_hi).
// c <= k: lane j takes candidate j.  c > k: k steps of a virtual Fisher-Yates shuffle of [0, c) -- the array is the
// identity except for <= k overrides, kept one per lane (position, value) and searched by ballot -- so lane i ends up
// with the i-th element of a uniformly random k-permutation.  All lanes run the same scalar recurrence.
__global__ __launch_bounds__(256) void uniform_lookup_kernel(const UniformArgs a) {
  const int lane = lane_id();
  const long long waves_total = (long long)gridDim.x * (blockDim.x >> 6);
  const int k = a.k;
  for (long long s = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < a.S; s += waves_total) {
    const int n = a.seeds[s];
    const bool live = n >= 0 && n < a.N;
    if (lane == 0 && (n >= a.N || n < -1 || (n == -1 && !a.allow_pad))) atomicOr(a.status, TGMX_ST_SEED_RANGE);
    long long ra = 0, c = 0;
    if (live) {
      ra = a.indptr[n];
      c = wave_prefix_count(a.recs, ra, a.indptr[n + 1], a.ev_hi, lane);
    }
    long long pick = lane;  // candidate index of output slot `lane`
    if (c > k) {
      const unsigned long long base = splitmix64(a.rng_seed ^ splitmix64(a.rng_stream ^ ((unsigned long long)(unsigned)n << 32)));
      long long ov_pos = -1, ov_val = 0;  // this lane's override of the virtual array
      int n_ov = 0;
      for (int i = 0; i < k; ++i) {
        const unsigned long long r = splitmix64(base + (unsigned long long)i);
        const long long j = i + (long long)__umul64hi(r, (unsigned long long)(c - i));  // uniform in [i, c)
        // a[j] and a[i] before the swap
        const unsigned long long mj = __ballot(ov_pos == j);
        const long long vj = mj ? __shfl(ov_val, __ffsll((long long)mj) - 1) : j;
        const unsigned long long mi = __ballot(ov_pos == i);
        const long long vi = mi ? __shfl(ov_val, __ffsll((long long)mi) - 1) : (long long)i;
        if (lane == i) pick = vj;  // output slot i
        if (j != i) {              // a[j] = old a[i] (a[i] itself is never read again)
          if (mj) {
            if (ov_pos == j) ov_val = vi;
          } else {
            if (lane == n_ov) {
              ov_pos = j;
              ov_val = vi;
            }
            ++n_ov;
          }
        }
      }
    }
    const long long m = c < k ? c : k;  // valid slots
    // ---- ids / times: lane j < k owns slot j ----
    int eid = -1;
    if (lane < k) {
      int nid = -1;
      long long ts = 0;
      if (lane < m) {
        const Rec r = a.recs[ra + pick];
        nid = r.nbr;
        ts = r.ts;
        eid = r.eid;
      }
      a.out_nid[s * k + lane] = nid;
      a.out_ts[s * k + lane] = ts;
    }
    // ---- features: the wave copies slot after slot (rows of edge_x addressed by eid) ----
    if (a.D > 0) {
      float* __restrict__ o = a.out_x + s * (long long)k * a.D;
      for (int j = 0; j < k; ++j) {
        const int e = __shfl(eid, j);
        if (e >= 0) {
          const float* __restrict__ x = a.edge_x + (long long)e * a.D;
          for (int col = lane; col < a.D; col += kWave) o[(long long)j * a.D + col] = x[col];
        } else {
          for (int col = lane; col < a.D; col += kWave) o[(long long)j * a.D + col] = 0.f;
        }
      }
    }
  }
}

}  // namespace tgmx

extern "C" int tgmx_uniform_lookup_csr(const int64_t* indptr, const tgmx_adj_t* adj, const float* edge_x, int32_t D,
                                       const int32_t* seeds, int64_t S, int32_t k, int64_t ev_hi, int32_t num_nodes,
                                       int32_t allow_pad, uint64_t rng_seed, uint64_t rng_stream, int32_t* out_nid,
                                       int64_t* out_ts, float* out_x, int32_t* status, tgmx_stream_t stream) {
  TGMX_REQUIRE(S >= 0 && k > 0 && k <= 64 && D >= 0 && num_nodes > 0 && ev_hi >= 0, "uniform_lookup: bad sizes S=%lld k=%d D=%d N=%d",
               (long long)S, k, D, num_nodes);
  if (S == 0) return TGMX_OK;
  TGMX_REQUIRE(indptr && adj && seeds && out_nid && out_ts && status, "uniform_lookup: null pointer");
  TGMX_REQUIRE(D == 0 || (edge_x && out_x), "uniform_lookup: D=%d but edge_x/out_x is null", D);
  TGMX_REQUIRE(((uintptr_t)adj & 15) == 0, "uniform_lookup: adj must be 16-byte aligned");
  UniformArgs a{indptr, reinterpret_cast<const Rec*>(adj), edge_x, seeds, out_nid, out_ts, out_x, status, S, ev_hi,
                (unsigned long long)rng_seed, (unsigned long long)rng_stream, D, k, num_nodes, allow_pad};
  long long blocks = (S + 3) / 4;
  if (blocks > (1 << 20)) blocks = 1 << 20;
  hipLaunchKernelGGL(uniform_lookup_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  TGMX_CHECK_LAUNCH("uniform_lookup");
  return TGMX_OK;
}

extern "C" int tgmx_recency_lookup_csr(const int64_t* indptr, const tgmx_adj_t* adj, const float* edge_x, int32_t D,
                                       const int32_t* seeds, const int64_t* qtimes, int64_t S, int32_t k, int32_t B,
                                       int64_t ev_lo, int64_t ev_hi, int32_t num_nodes, int32_t allow_pad,
                                       int32_t* out_nid, int64_t* out_ts, float* out_x, int32_t* status,
                                       tgmx_stream_t stream, tgmx_event_t ev_start, tgmx_event_t ev_stop) {
  TGMX_REQUIRE(S >= 0 && k > 0 && B >= k && D >= 0 && num_nodes > 0, "recency_lookup_csr: bad sizes S=%lld k=%d B=%d D=%d N=%d",
               (long long)S, k, B, D, num_nodes);
  if (S == 0) return TGMX_OK;
  TGMX_REQUIRE(indptr && adj && seeds && qtimes && out_nid && out_ts && status, "recency_lookup_csr: null pointer");
  TGMX_REQUIRE(D == 0 || (edge_x && out_x), "recency_lookup_csr: D=%d but edge_x/out_x is null", D);
  TGMX_REQUIRE(((uintptr_t)adj & 15) == 0, "recency_lookup_csr: adj must be 16-byte aligned");
  LookupArgs a{};
  a.indptr = indptr; a.recs = reinterpret_cast<const Rec*>(adj); a.write_pos = nullptr; a.edge_x = edge_x;
  a.seeds = seeds; a.qtimes = qtimes; a.out_nid = out_nid; a.out_ts = out_ts; a.out_x = out_x; a.status = status;
  a.S = S; a.ev_lo = ev_lo; a.ev_hi = ev_hi; a.D = D; a.k = k; a.B = B; a.N = num_nodes; a.allow_pad = allow_pad;
  return launch_lookup<false>(a, (hipStream_t)stream, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}

extern "C" int tgmx_ring_lookup(const tgmx_adj_t* ring, const int32_t* write_pos, const float* ring_x, int32_t D,
                                const int32_t* seeds, const int64_t* qtimes, int64_t S, int32_t k, int32_t B,
                                int32_t num_nodes, int32_t allow_pad, int32_t* out_nid, int64_t* out_ts, float* out_x,
                                int32_t* status, tgmx_stream_t stream, tgmx_event_t ev_start, tgmx_event_t ev_stop) {
  TGMX_REQUIRE(S >= 0 && k > 0 && B >= k && D >= 0 && num_nodes > 0, "ring_lookup: bad sizes S=%lld k=%d B=%d D=%d N=%d",
               (long long)S, k, B, D, num_nodes);
  if (S == 0) return TGMX_OK;
  TGMX_REQUIRE(ring && write_pos && seeds && qtimes && out_nid && out_ts && status, "ring_lookup: null pointer");
  TGMX_REQUIRE(D == 0 || (ring_x && out_x), "ring_lookup: D=%d but ring_x/out_x is null", D);
  TGMX_REQUIRE(((uintptr_t)ring & 15) == 0, "ring_lookup: ring must be 16-byte aligned");
  TGMX_REQUIRE((long long)B * num_nodes < 2147483647LL, "ring_lookup: num_nodes*B overflows int32");
  LookupArgs a{};
  a.indptr = nullptr; a.recs = reinterpret_cast<const Rec*>(ring); a.write_pos = write_pos; a.edge_x = ring_x;
  a.seeds = seeds; a.qtimes = qtimes; a.out_nid = out_nid; a.out_ts = out_ts; a.out_x = out_x; a.status = status;
  a.S = S; a.ev_lo = 0; a.ev_hi = 0; a.D = D; a.k = k; a.B = B; a.N = num_nodes; a.allow_pad = allow_pad;
  return launch_lookup<true>(a, (hipStream_t)stream, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}

constexpr int kScratchHead = 1024;  // ints (4 KiB, zero at first use: barrier words + the tail commit's grouped counters; the layouts behind it stay 256-byte aligned)

static int fill_update_args(UpdateArgs& a, tgmx_adj_t* ring, int32_t* write_pos, float* ring_x, int32_t D, int32_t B,
                            int32_t num_nodes, const int32_t* src, const int32_t* dst, const int64_t* ts,
                            const float* edge_x, int64_t n, int64_t eid0, int32_t directed, int32_t key_wrap32,
                            int32_t* scratch, int32_t* status) {
  TGMX_REQUIRE(n > 0 && B > 0 && num_nodes > 0 && D >= 0, "ring_update: bad sizes n=%lld B=%d N=%d D=%d", (long long)n, B,
               num_nodes, D);
  TGMX_REQUIRE(ring && write_pos && src && dst && ts && scratch && status, "ring_update: null pointer");
  TGMX_REQUIRE(D == 0 || ring_x, "ring_update: D=%d but ring_x is null", D);
  TGMX_REQUIRE(eid0 < 0 || eid0 + n <= 2147483647LL, "ring_update: edge ids overflow int32");
  TGMX_REQUIRE((long long)B * num_nodes < 2147483647LL, "ring_update: num_nodes*B overflows int32");
  TGMX_REQUIRE(2 * n < 2147483647LL, "ring_update: batch too large");
  TGMX_REQUIRE(((uintptr_t)scratch & 255) == 0, "ring_update: scratch must be 256-byte aligned");
  a = UpdateArgs{};
  a.ring = reinterpret_cast<Rec*>(ring); a.write_pos = write_pos; a.ring_x = ring_x; a.edge_x = edge_x;
  a.src = src; a.dst = dst; a.ts = ts; a.status = status;
  a.n = n; a.m = directed ? n : 2 * n; a.eid0 = eid0; a.B = B; a.N = num_nodes; a.D = D; a.key_wrap32 = key_wrap32;
  a.barrier = scratch;      // first kScratchHead ints: the riders' barrier words (zero when the buffer is first used)
  scratch += kScratchHead;  // everything else lives behind them, in every path
  a.sorted_j = scratch; a.sorted_node = scratch + a.m; a.target = scratch + 2 * a.m; a.winner = scratch + 3 * a.m;
  return TGMX_OK;
}

// scratch layout of the large-batch path (byte offsets from a 256-byte aligned base)
struct LargeScratch {
  size_t span, keys_in, keys_out, vals_in, hash_key, hash_maxp, run_start, run_len, temp, temp_bytes, total;
  int hash_bits;
};

static int large_scratch_layout(long long m, LargeScratch& w) {
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t off = up((size_t)4 * m * sizeof(int32_t));  // sorted_j, sorted_node, target, winner
  w.span = off; off = up(off + 16);
  w.keys_in = off; off = up(off + (size_t)m * 8);
  w.keys_out = off; off = up(off + (size_t)m * 8);
  w.vals_in = off; off = up(off + (size_t)m * 4);
  w.hash_bits = 8;
  while ((1ll << w.hash_bits) < 2 * m) ++w.hash_bits;  // load factor <= 0.5
  w.hash_key = off; off = up(off + ((size_t)4 << w.hash_bits));
  w.hash_maxp = off; off = up(off + ((size_t)4 << w.hash_bits));
  w.run_start = off; off = up(off + (size_t)m * 4);
  w.run_len = off; off = up(off + (size_t)m * 4);
  size_t tb = 0;
  const hipError_t err = rocprim::radix_sort_pairs(nullptr, tb, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                   (const unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)m, 0u, 64u);
  if (err != hipSuccess) {
    set_error("ring_update: radix sort workspace query failed: %s", hipGetErrorString(err));
    return TGMX_E_LAUNCH;
  }
  size_t sb = 0;
  const hipError_t err2 = rocprim::inclusive_scan(nullptr, sb, rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), RunFlag{nullptr}),
                                                  (int*)nullptr, (size_t)m, rocprim::maximum<int>());
  if (err2 != hipSuccess) {
    set_error("ring_update: scan workspace query failed: %s", hipGetErrorString(err2));
    return TGMX_E_LAUNCH;
  }
  w.temp = off;
  w.temp_bytes = tb > sb ? tb : sb;
  w.total = up(off + w.temp_bytes) + 256;
  return TGMX_OK;
}

// front: everything up to the placement decisions -- nothing there writes ring state, so it may run next to the lookups
// of the same batch on another stream; back: the writes (ring records, write_pos, feature rows)
static int launch_update_large_front(UpdateArgs& a, int32_t* scratch, hipStream_t st, bool beside_lookups = false) {
  LargeScratch w;
  const int rc = large_scratch_layout(a.m, w);
  if (rc) return rc;
  char* base = reinterpret_cast<char*>(scratch);  // the caller's buffer is 256-byte aligned (checked by the entry points)
  a.span = reinterpret_cast<long long*>(base + w.span);
  a.keys_in = reinterpret_cast<unsigned long long*>(base + w.keys_in);
  auto* keys_out = reinterpret_cast<unsigned long long*>(base + w.keys_out);
  a.vals_in = reinterpret_cast<unsigned int*>(base + w.vals_in);
  a.hash_key = reinterpret_cast<int32_t*>(base + w.hash_key);
  a.hash_maxp = reinterpret_cast<int32_t*>(base + w.hash_maxp);
  a.hash_bits = w.hash_bits;
  a.run_start = reinterpret_cast<int32_t*>(base + w.run_start);
  a.run_len = reinterpret_cast<int32_t*>(base + w.run_len);
  const unsigned blocks = (unsigned)((a.m + 255) / 256);
  static const bool no_front_block = getenv("TGMX_NO_FRONT_BLOCK") != nullptr;  // A/B knob: the chain of small launches
  // (on an idle chip the chain is the faster of the two: 74 vs 105 us for the whole update at m = 8192 -- one workgroup is one CU)
  const bool front_block = beside_lookups && a.m <= kFrontMaxM && !no_front_block;
  if (!front_block) {
    (void)hipMemsetAsync(a.hash_key, 0xFF, (size_t)8 << w.hash_bits, st);  // keys and max positions (adjacent): all -1
    if (!a.sorted_ts) {
      (void)hipMemsetAsync(a.span, 0x80, sizeof(long long), st);  // 0x8080...: far below any timestamp
      const unsigned span_blocks = (unsigned)((a.n + 2047) / 2048 < 64 ? (a.n + 2047) / 2048 : 64);
      hipLaunchKernelGGL(ring_update_span_kernel, dim3(span_blocks), dim3(256), 0, st, a);
    }
  }
  // how many key bits can be set?  (only with the caller's promise 0 <= t <= ts_bound)
  a.sort_bits = 64;
  if (a.ts_bound > 0) {
    int bits = 64;
    if (a.key_wrap32) {
      if (a.ts_bound < (1ll << 61)) {
        const unsigned long long top = (1ull << 32) + (unsigned long long)a.ts_bound;  // key + 2^31 < 2^32 + ts_bound
        bits = 64 - __builtin_clzll(top);
      }
    } else {
      const __int128 top = (__int128)a.N * ((__int128)a.ts_bound + 1);  // node * span + t < N * (ts_bound + 1)
      if (top < ((__int128)1 << 62)) bits = 64 - __builtin_clzll((unsigned long long)top);
    }
    a.sort_bits = bits;
  }
  if (front_block) {  // one workgroup: keys, sort, scatter, runs -- then the placement (folded into the one workgroup, its
    // dependent atomics under the lookups' load took longer than the lookup launch: measured, step 154 -> 189 us)
    hipLaunchKernelGGL(ring_update_front_block_kernel, dim3(1), dim3(kBlockThreads), 0, st, a);
    hipLaunchKernelGGL(ring_update_place_kernel, dim3(blocks), dim3(256), 0, st, a);
    return TGMX_OK;
  }
  hipLaunchKernelGGL(ring_update_keys_kernel, dim3(blocks), dim3(256), 0, st, a);
  size_t tb = w.temp_bytes;
  const hipError_t err = rocprim::radix_sort_pairs(base + w.temp, tb, (const unsigned long long*)a.keys_in, keys_out,
                                                   (const unsigned int*)a.vals_in, reinterpret_cast<unsigned int*>(a.sorted_j),
                                                   (size_t)a.m, 0u, (unsigned)a.sort_bits, st);
  if (err != hipSuccess) {
    set_error("ring_update: radix sort failed: %s", hipGetErrorString(err));
    return TGMX_E_LAUNCH;
  }
  hipLaunchKernelGGL(ring_update_scatter_kernel, dim3(blocks), dim3(256), 0, st, a);
  // run_start = inclusive max-scan of "p if p opens a run else 0", the flag evaluated inside the scan's load (no flags launch)
  tb = w.temp_bytes;
  const auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), RunFlag{a.sorted_node});
  const hipError_t err2 = rocprim::inclusive_scan(base + w.temp, tb, flags, a.run_start, (size_t)a.m, rocprim::maximum<int>(), st);
  if (err2 != hipSuccess) {
    set_error("ring_update: scan failed: %s", hipGetErrorString(err2));
    return TGMX_E_LAUNCH;
  }
  hipLaunchKernelGGL(ring_update_ends_kernel, dim3(blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(ring_update_place_kernel, dim3(blocks), dim3(256), 0, st, a);
  return TGMX_OK;
}

static void launch_update_large_back(const UpdateArgs& a, hipStream_t st) {
  if (a.D > 0) {
    hipLaunchKernelGGL(ring_update_write_feat_kernel, dim3((unsigned)((a.m + 3) / 4)), dim3(256), 0, st, a);
    return;
  }
  const unsigned blocks = (unsigned)((a.m + 255) / 256);
  hipLaunchKernelGGL(ring_update_write_kernel, dim3(blocks), dim3(256), 0, st, a);
}

static int launch_update_large(UpdateArgs& a, int32_t* scratch, hipStream_t st) {
  if (const int rc = launch_update_large_front(a, scratch, st)) return rc;
  launch_update_large_back(a, st);
  return TGMX_OK;
}

extern "C" size_t tgmx_ring_update_scratch_bytes(int64_t n, int32_t directed) {
  const long long m = directed ? n : 2 * n;
  constexpr size_t head = kScratchHead * sizeof(int32_t);
  if (m <= 0) return head + 256;
  if (m <= kBlockMaxM) return head + ((size_t)12 * m + 16) * sizeof(int32_t) + 256 + 12 * kChunk;  // + chunk padding of the spread sort
  LargeScratch w;
  if (large_scratch_layout(m, w)) return 0;
  return head + w.total;
}

extern "C" int tgmx_ring_update(tgmx_adj_t* ring, int32_t* write_pos, float* ring_x, int32_t D, int32_t B,
                                int32_t num_nodes, const int32_t* src, const int32_t* dst, const int64_t* ts,
                                const float* edge_x, int64_t n, int64_t eid0, int32_t directed, int32_t key_wrap32,
                                int32_t* scratch, int32_t* status, tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0, "ring_update: bad sizes n=%lld", (long long)n);
  if (n == 0) return TGMX_OK;
  UpdateArgs a;
  const int rc = fill_update_args(a, ring, write_pos, ring_x, D, B, num_nodes, src, dst, ts, edge_x, n, eid0, directed,
                                  key_wrap32, scratch, status);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (a.m <= kBlockMaxM) launch_update_block(a, scratch + kScratchHead, st);
  else if (const int rl = launch_update_large(a, scratch + kScratchHead, st)) return rl;
  TGMX_CHECK_LAUNCH("ring_update");
  return TGMX_OK;
}

// Large batches (m > kBlockMaxM: the rocPRIM path) inside tgmx_recency_step: sort, run analysis and placement
// decisions read no ring state that the lookups of the same batch change, so they run on a library-owned side stream
// next to the lookups; only the writes wait for both.  (For small batches the riders do the same inside the lookup
// launches without any stream traffic -- four extra runtime calls per batch only pay when the batch is big: the
// comment-shaped step is 220 us of kernels against 60 us of host.)
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
static SideStream* side_stream_for_current_device() {
  static SideStream side[64];
  static const bool off = getenv("TGMX_NO_SIDE_STREAM") != nullptr;  // A/B knob
  int dev = 0;
  if (off || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  SideStream& s = side[dev];
  if (!s.stream) {
    // highest priority the device offers: the chain of small launches on this stream runs beside a lookup launch that keeps the
    // whole chip busy, and every one of its launches would otherwise queue behind that launch's workgroups
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    static const bool no_prio = getenv("TGMX_SIDE_NO_PRIO") != nullptr;  // A/B knob
    if (hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, no_prio ? prio_lo : prio_hi) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) {
      s.stream = nullptr;
      return nullptr;
    }
  }
  return &s;
}

// Can hop 0 and hop 1 of this step be one launch?  (B <= 64; hop 1 not better served by the packed narrow-row kernel.)
static bool plan_fuse01(const tgmx_recency_step_t* s, long long S0) {
  static const bool no_fuse = getenv("TGMX_NO_FUSE") != nullptr;  // A/B knob
  static const bool no_ride = getenv("TGMX_NO_RIDE") != nullptr;
  if (no_fuse || s->n_hops < 2 || S0 <= 0 || s->B > kWave || s->k[0] <= 0 || s->k[1] <= 0) return false;
  // static index: a hop-1 wave would repeat hop 0's two prefix searches (dependent reads); that only pays when its own
  // gather is long (comment-shaped, D = 16: 26.9 -> 19.2 G sampled-edges/s fused; wiki-shaped, D = 172: 4.8 -> 5.1)
  if (s->indptr != nullptr && (long long)s->k[1] * s->D * 4 < 4096) return false;
  // edge features BY ID (out_eid set, no out_x): no feature row is copied, the packed narrow-row kernel does not apply (it does not
  // publish edge ids) and the two hops would be two wave-per-seed launches of ~12 us each at the review shape: one launch instead
  static const bool fuse_by_id = !(getenv("TGMX_FUSE_BY_ID") && atoi(getenv("TGMX_FUSE_BY_ID")) == 0);  // A/B knob
  const bool by_id = s->out_eid[1] != nullptr && s->out_x[1] == nullptr;
  if (s->indptr == nullptr && !(by_id && fuse_by_id)) {  // streaming rings: the packed kernel takes narrow hop-1 rows
    LookupArgs a{};
    a.B = s->B; a.D = s->D; a.edge_x = s->ring_x; a.out_x = s->out_x[0];
    if (prepare_lookup(a, s->out_x[1], s->k[1]) < 0) return false;
    if (packed_group_lanes(a, s->k[1], true) < 64) return false;  // (review-shaped, forced through the fused launch: 38.7 vs 28.9 us/step)
  }
  const long long m = s->directed ? s->n : 2 * s->n;
  (void)no_ride;
  (void)m;
  return true;
}

extern "C" int tgmx_recency_step_plan(const tgmx_recency_step_t* s) {
  if (!s) return 0;
  long long S0 = s->S0;
  if (s->n_groups > 0) {
    S0 = 0;
    for (int g = 0; g < s->n_groups && g < TGMX_MAX_SEED_GROUPS; ++g) S0 += s->grp_n[g];
  }
  return plan_fuse01(s, S0) ? 1 : 0;
}

extern "C" int tgmx_recency_step(const tgmx_recency_step_t* s, tgmx_stream_t stream) {
  TGMX_REQUIRE(s, "recency_step: null argument block");
  TGMX_REQUIRE(s->n_groups >= 0 && s->n_groups <= TGMX_MAX_SEED_GROUPS && s->n_hops >= 0 && s->n_hops <= TGMX_MAX_HOPS,
               "recency_step: n_groups=%d n_hops=%d out of range", s->n_groups, s->n_hops);
  TGMX_REQUIRE(s->B > 0 && s->num_nodes > 0 && s->D >= 0 && s->n >= 0, "recency_step: bad sizes B=%d N=%d D=%d n=%lld", s->B,
               s->num_nodes, s->D, (long long)s->n);
  const bool csr = s->indptr != nullptr;
  TGMX_REQUIRE(s->ring && (csr || s->write_pos) && s->status, "recency_step: null state pointer");
  TGMX_REQUIRE(!csr || s->n == 0, "recency_step: the static index takes no update (n must be 0)");
  TGMX_REQUIRE(((uintptr_t)s->ring & 15) == 0, "recency_step: ring must be 16-byte aligned");
  TGMX_REQUIRE(csr || (long long)s->B * s->num_nodes < 2147483647LL, "recency_step: num_nodes*B overflows int32");
  hipStream_t st = (hipStream_t)stream;

  // ---- hop-0 seeds: groups are concatenated by the hop-0 lookup itself (or by nothing when there is no hop)
  SeedGroups grp{};
  grp.gen = -1;
  long long S = s->S0;
  if (s->n_groups > 0) {
    S = 0;
    const int gen = s->neg_out ? s->neg_group : -1;  // negatives generated in place of seed group neg_group
    TGMX_REQUIRE(gen < s->n_groups, "recency_step: neg_group=%d but %d seed groups", gen, s->n_groups);
    if (gen >= 0) {
      TGMX_REQUIRE(s->neg_low < s->neg_high && s->neg_time_out, "recency_step: generated negatives need low < high and neg_time_out");
      grp.gen = gen; grp.gen_low = s->neg_low; grp.gen_range = (unsigned)((long long)s->neg_high - s->neg_low);
      grp.gen_seed = s->neg_seed; grp.gen_call = s->neg_call; grp.gen_index0 = (unsigned long long)s->neg_index0; grp.gen_out = s->neg_out; grp.gen_out_ts = s->neg_time_out;
    }
    for (int g = 0; g < s->n_groups; ++g) {
      TGMX_REQUIRE(s->grp_n[g] >= 0 && (s->grp_n[g] == 0 || ((g == gen || s->grp_nid[g]) && s->grp_ts[g])), "recency_step: seed group %d", g);
      S += s->grp_n[g];
      grp.nid[g] = s->grp_nid[g]; grp.ts[g] = s->grp_ts[g]; grp.end[g] = S;
    }
    TGMX_REQUIRE(S == 0 || (s->seed_nid0 && s->seed_ts0), "recency_step: null hop-0 seed output");
    TGMX_REQUIRE(S == 0 || s->n_hops > 0, "recency_step: seed groups need at least one hop");
    grp.out_nid = s->seed_nid0; grp.out_ts = s->seed_ts0; grp.groups = s->n_groups;
  }

  // ---- ring update, front half: batches of up to kBlockMaxM entries are sorted by workgroups riding along with the
  // lookup launches (chunk sort with hop 0, merge with hop 1); only the placement runs after the lookups
  UpdateArgs u{};
  unsigned side_chunks = 0;
  bool ride_place = false;  // m <= 1024: hop 1 carries the merge AND the placement decisions
  if (s->n > 0) {
    const int rc = fill_update_args(u, s->ring, s->write_pos, s->ring_x, s->D, s->B, s->num_nodes, s->src, s->dst, s->ts,
                                    s->edge_x, s->n, s->eid0, s->directed, s->key_wrap32, s->scratch, s->status);
    if (rc) return rc;
    u.ts_bound = s->ts_bound;
    u.guard_mask = s->guard_seed_errors ? (TGMX_ST_SEED_RANGE | TGMX_ST_SEED_TIME) : 0;
    u.sorted_ts = s->sorted_ts;
    static const bool no_ride = getenv("TGMX_NO_RIDE") != nullptr;  // A/B knob: the update as its own launches
    if (u.m <= kBlockMaxM && s->n_hops > 0 && S > 0 && !no_ride) side_chunks = set_chunk_scratch(u, s->scratch + kScratchHead);
    ride_place = side_chunks > 0 && u.m <= kRidePlaceMaxM && s->n_hops >= 2;
  }
  // 1024 < m <= 4096 (the replicated update of a 4- / 8-rank global wiki batch): the front half -- sort, merge, placement
  // decisions -- CAN run on the side stream next to the lookups instead of riders + a placement launch behind them.
  // Measured on MI355X and rejected: the fork / join dependency between the streams costs more than the placement launch it
  // hides (wiki, rank 7 of 8: 84.1 vs 60.4 us per step; rank 3 of 4: 70.4 vs 53.2).  Off unless TGMX_SIDE_MID=1 (A/B knob).
  static const bool no_side_mid = getenv("TGMX_SIDE_MID") == nullptr;
  SideStream* mid = nullptr;
  if (side_chunks > 0 && !ride_place && u.m > kRidePlaceMaxM && !no_side_mid && (mid = side_stream_for_current_device()) != nullptr) {
    (void)hipEventRecord(mid->fork, st);  // the batch's inputs and the previous batch's ring writes are complete
    (void)hipStreamWaitEvent(mid->stream, mid->fork, 0);
    launch_update_mid_front(u, side_chunks, mid->stream);
    (void)hipEventRecord(mid->join, mid->stream);
    side_chunks = 0;  // no riders in the lookup launches
  }
  // m <= 1024 and the placement rides: the write-back (records, write_pos, feature rows) CAN run as the tail of the last
  // lookup launch -- its last ceil(m / 4) workgroups prefetch their decisions, wait for the others, then store -- instead
  // of a launch of its own.  Measured on MI355X (wiki shape): lookup 37.9 + commit 5.3 us as two launches, 44.9 us as one
  // (the launch-wide wait costs more than the launch boundary it replaces), so it is OFF unless TGMX_TAIL=1 (A/B knob;
  // results identical, covered by the same tests).
  static const bool use_tail = getenv("TGMX_TAIL") != nullptr;
  const unsigned tail_blocks = (ride_place && use_tail && s->n_hops == 2) ? (unsigned)((u.m + 3) / 4) : 0u;
  SideStream* side = nullptr;  // set: the large update's front half runs on the side stream next to the lookups
  bool side_pending = false;   // its launches are enqueued BEHIND the first lookup launch (below)
  if (s->n > 0 && u.m > kBlockMaxM && s->n_hops > 0 && S > 0 && (side = side_stream_for_current_device()) != nullptr) {
    (void)hipEventRecord(side->fork, st);  // the batch's inputs and the previous batch's ring writes are complete
    (void)hipStreamWaitEvent(side->stream, side->fork, 0);
    side_pending = true;
  }
  // The front half is a dozen small launches (~40 us of host time).  Enqueued in front of the lookups, the caller's stream has
  // nothing to run until the host gets to the first lookup launch; behind it, that launch covers the host's work.
  auto enqueue_side = [&]() -> int {
    if (!side_pending) return TGMX_OK;
    side_pending = false;
    // "beside a saturating lookup launch": only then do the chain's small kernels crawl (and the one-workgroup front half pay); beside
    // a few thousand seeds -- an 8-rank share of a 4096-edge batch -- the chain is the faster of the two and the step waits for it
    long long widest = S;
    for (int h = 0; h + 1 < s->n_hops; ++h) widest *= s->k[h];
    const bool saturating = widest >= 2ll * kWave * device_cu_count();
    if (const int rl = launch_update_large_front(u, s->scratch + kScratchHead, side->stream, saturating)) return rl;
    (void)hipEventRecord(side->join, side->stream);
    return TGMX_OK;
  };

  // ---- lookups, hop by hop (hop h + 1 consumes hop h's outputs in place); hops 0 and 1 as one launch when possible
  const int32_t* cur_n = s->seed_nid0;
  const int64_t* cur_t = s->seed_ts0;
  int h = 0;
  if (plan_fuse01(s, S)) {
    const int k0 = s->k[0], k1 = s->k[1];
    TGMX_REQUIRE(s->B >= k0 && s->B >= k1, "recency_step: k=[%d, %d] but B=%d", k0, k1, s->B);
    TGMX_REQUIRE(cur_n && cur_t && s->out_nid[0] && s->out_ts[0] && s->out_nid[1] && s->out_ts[1] &&
                     (s->D == 0 || (s->ring_x && (s->out_x[0] || s->out_eid[0]) && (s->out_x[1] || s->out_eid[1]))), "recency_step: null pointer at hop 0 / 1");
    LookupArgs a{};
    a.grp = grp;
    a.indptr = s->indptr; a.recs = reinterpret_cast<const Rec*>(s->ring); a.write_pos = s->write_pos; a.edge_x = s->ring_x;
    a.seeds = cur_n; a.qtimes = cur_t; a.out_nid = s->out_nid[0]; a.out_ts = s->out_ts[0]; a.out_x = s->out_x[0];
    a.k1 = k1; a.out_nid1 = s->out_nid[1]; a.out_ts1 = s->out_ts[1]; a.out_x1 = s->out_x[1];
    a.out_valid = s->out_valid[0]; a.out_valid1 = s->out_valid[1];
    a.out_valid_prev = s->out_valid_prev[0]; a.out_valid_prev1 = s->out_valid_prev[1];
    a.out_eid = s->out_eid[0]; a.out_eid1 = s->out_eid[1];
    a.status = s->status; a.S = S; a.D = s->D; a.k = k0; a.B = s->B; a.N = s->num_nodes; a.allow_pad = 0;
    a.ev_lo = s->ev_lo; a.ev_hi = s->ev_hi;
    a.x_by_pos = csr && s->csr_x_by_pos;
    const bool timed = s->timed_hop == 0 || s->timed_hop == 1;
    hipEvent_t e0 = timed ? (hipEvent_t)s->ev_start : nullptr, e1 = timed ? (hipEvent_t)s->ev_stop : nullptr;
    // riders: m <= 512 -> one workgroup does it all and only the commit follows; 512 < m <= 1024 -> a rider per chunk, the last one
    // out places (TGMX_THREE_PHASE=0: one workgroup, A/B); else sort | barrier | merge and the placement is its own launch
    static const bool three_phase_on = !(getenv("TGMX_THREE_PHASE") && atoi(getenv("TGMX_THREE_PHASE")) == 0);
    const bool three_phase = three_phase_on && ride_place && u.m > 512 && !tail_blocks;
    const int rc = csr ? launch_fused01<false>(a, st, e0, e1, nullptr, 0, 0)
                       : launch_fused01<true>(a, st, e0, e1, side_chunks > 0 ? &u : nullptr,
                                              ride_place ? (three_phase ? kSideSortMergePlace : kSideAll) : kSideSortMerge,
                                              (ride_place && !three_phase) ? 1u : side_chunks, tail_blocks);
    if (rc) return rc;
    if (const int rs = enqueue_side()) return rs;
    cur_n = s->out_nid[1];
    cur_t = s->out_ts[1];
    S *= (long long)k0 * k1;
    h = 2;
  }
  for (; h < s->n_hops && S > 0; ++h) {
    const int k = s->k[h];
    TGMX_REQUIRE(k > 0 && s->B >= k, "recency_step: hop %d has k=%d, B=%d", h, k, s->B);
    TGMX_REQUIRE(cur_n && cur_t && s->out_nid[h] && s->out_ts[h] && (s->D == 0 || (s->ring_x && (s->out_x[h] || s->out_eid[h]))),
                 "recency_step: null pointer at hop %d", h);
    LookupArgs a{};
    if (h == 0) a.grp = grp;
    a.indptr = s->indptr; a.recs = reinterpret_cast<const Rec*>(s->ring); a.write_pos = s->write_pos; a.edge_x = s->ring_x;
    a.seeds = cur_n; a.qtimes = cur_t; a.out_nid = s->out_nid[h]; a.out_ts = s->out_ts[h]; a.out_x = s->out_x[h];
    a.out_valid = s->out_valid[h]; a.out_valid_prev = s->out_valid_prev[h];
    a.out_eid = s->out_eid[h];
    a.status = s->status; a.S = S; a.D = s->D; a.k = k; a.B = s->B; a.N = s->num_nodes; a.allow_pad = h > 0;
    a.ev_lo = s->ev_lo; a.ev_hi = s->ev_hi;
    a.cursor = csr ? reinterpret_cast<long long*>(s->csr_cursor) : nullptr;
    a.x_by_pos = csr && s->csr_x_by_pos;
    a.leave_room = side != nullptr;
    const bool timed = h == s->timed_hop;
    hipEvent_t e0 = timed ? (hipEvent_t)s->ev_start : nullptr, e1 = timed ? (hipEvent_t)s->ev_stop : nullptr;
    const bool ride = side_chunks > 0 && h < 2;
    // m <= 1024, two lookup launches: the merge rides hop 0 with the chunk sorts (its riders meet at a barrier), hop 1's single rider
    // only decides the placement -- the one workgroup that merged AND placed made hop 1 last 17.7 us at the review shape (TGMX_SPLIT_MERGE=0: A/B)
    static const bool split_merge_on = !(getenv("TGMX_SPLIT_MERGE") && atoi(getenv("TGMX_SPLIT_MERGE")) == 0);
    const bool split_merge = split_merge_on && ride_place && s->n_hops >= 2;
    // round 6: hop 0 carries the chunk sorts only, hop 1 the merge (a rider per chunk) + the placement (the last rider out): the riders no
    // longer pace the first launch (TGMX_MERGE_LATE=0: sort + barrier + merge in hop 0, placement in hop 1, as in rounds 3-5 -- A/B)
    static const bool merge_late_on = !(getenv("TGMX_MERGE_LATE") && atoi(getenv("TGMX_MERGE_LATE")) == 0);
    const bool merge_late = merge_late_on && split_merge && side_chunks > 1;
    const int stage = h == 0 ? ((split_merge && !merge_late) ? kSideSortMerge : kSideSort)
                             : (ride_place ? (merge_late ? kSideMergePlace : (split_merge ? kSidePlaceOnly : kSidePlace)) : kSideMerge);
    const int rc = csr ? launch_lookup<false>(a, st, e0, e1)
                       : launch_lookup<true>(a, st, e0, e1, ride ? &u : nullptr, stage,
                                             (h == 1 && ride_place && !merge_late) ? 1u : side_chunks, (h == 1 && ride_place) ? tail_blocks : 0u);
    if (rc) return rc;
    if (const int rs = enqueue_side()) return rs;
    cur_n = s->out_nid[h];
    cur_t = s->out_ts[h];
    S *= k;
  }
  if (const int rs = enqueue_side()) return rs;

  // ---- ring update (after every lookup, recency.py:161-163)
  if (s->n > 0) {
    if (mid) {
      (void)hipStreamWaitEvent(st, mid->join, 0);
      hipLaunchKernelGGL(ring_update_feat_kernel<true>, dim3((unsigned)((u.m + 3) / 4)), dim3(256), 0, st, u);
    } else if (ride_place && tail_blocks) {
      // written by the tail of the last lookup launch
    } else if (ride_place) {
      hipLaunchKernelGGL(ring_update_feat_kernel<true>, dim3((unsigned)((u.m + 3) / 4)), dim3(256), 0, st, u);
    } else if (side_chunks > 0) {
      if (s->n_hops < 2) hipLaunchKernelGGL(ring_update_merge_kernel, dim3(side_chunks), dim3(kChunk), 0, st, u);
      launch_update_presorted(u, st);
    } else if (u.m <= kBlockMaxM) {
      launch_update_block(u, s->scratch + kScratchHead, st);
    } else if (side) {
      (void)hipStreamWaitEvent(st, side->join, 0);
      launch_update_large_back(u, st);
    } else if (const int rl = launch_update_large(u, s->scratch + kScratchHead, st)) {
      return rl;
    }
  }
  TGMX_CHECK_LAUNCH("recency_step");
  return TGMX_OK;
}

extern "C" int tgmx_ring_reset(tgmx_adj_t* ring, int32_t* write_pos, int32_t B, int32_t num_nodes,
                               tgmx_stream_t stream) {
  TGMX_REQUIRE(ring && write_pos && B > 0 && num_nodes > 0, "ring_reset: bad arguments");
  const long long nrec = (long long)B * num_nodes;
  long long blocks = (nrec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(ring_reset_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<Rec*>(ring), write_pos, nrec, num_nodes);
  TGMX_CHECK_LAUNCH("ring_reset");
  return TGMX_OK;
}
