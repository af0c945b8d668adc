// TGN memory module + graph-attention embedding (fp32, forward) for gfx950.
//
// Reference: tgm/nn/encoder/tgn.py.  The reference keeps a
// Python dict  node -> (src[], dst[], t[], raw_msg[])  per role and rebuilds it with
// .tolist() loops every batch (tgn.py:218-243).  Here a node's stored events are a
// (lo, cnt) window into an append-only device log (other id, t, raw message row),
// written once per batch in node-sorted order; messages are never materialised for
// all events -- the aggregation kernel walks a node's window and builds only the
// row(s) it needs.  Everything is a segmented, HBM/latency-bound reduction: one wave
// per node row, lanes across the message columns.  The dense GRU contractions reuse
// sgemm_nt (exact-fp32 MFMA) from tgat.hip.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"
#include "lanes.h"

namespace tgmx {

// ---------------------------------------------------------------------------
// Message store update (tgn.py:218-229): entries are given in node-sorted (stable) order.
//   p: sorted position, e = perm[p] the batch entry it came from
//   log[base + p] = (other[e], t[e], raw[e, :]);  store[node] = (base + left[p], right[p] - left[p])
// ---------------------------------------------------------------------------
struct StoreArgs {
  const int64_t* perm;      // [n]
  const int32_t* node_sorted;  // [n]
  const int64_t* left;      // [n] first sorted position of this node's run
  const int64_t* right;     // [n] one past the last
  const int32_t* other;     // [n] (batch order)
  const int64_t* t;         // [n]
  const float* raw;         // [n, D]
  int32_t* log_other;
  int64_t* log_t;
  float* log_raw;
  int64_t* st_lo;           // [N]
  int32_t* st_cnt;          // [N]
  long long n, base;
  int D;
};

__global__ __launch_bounds__(256) void tgn_store_kernel(const StoreArgs a) {
  const long long p = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per entry
  if (p >= a.n) return;
  const int lane = lane_id();
  const long long e = a.perm[p];
  const long long dst = a.base + p;
  if (lane == 0) {
    a.log_other[dst] = a.other[e];
    a.log_t[dst] = a.t[e];
    const int node = a.node_sorted[p];
    a.st_lo[node] = a.base + a.left[p];  // every entry of the run writes the same pair
    a.st_cnt[node] = (int)(a.right[p] - a.left[p]);
  }
  for (int c = lane; c < a.D; c += kWave) a.log_raw[dst * a.D + c] = a.raw[e * a.D + c];
}

// ---------------------------------------------------------------------------
// Aggregation (tgn.py:191-209, 43-63, 66-74): per row (node v) walk the source-role window then the
// destination-role window.
//   message(event) = [mem[v] | mem[other] | raw | cos(fma(float(t - last_update[v]), w, b))]
//   LAST: the message of the event with the largest float32(t), first one in walk order on ties
//   MEAN: arithmetic mean of the messages
//   new_last_update = max t (int64) over the events, 0 without events; aggr = 0 without events
// ---------------------------------------------------------------------------
struct AggrArgs {
  const int32_t* nodes;     // [R]
  const float* memory;      // [N, M]
  const int64_t* last_update;  // [N]
  const int64_t* st_lo[2];
  const int32_t* st_cnt[2];
  const int32_t* log_other;
  const int64_t* log_t;
  const float* log_raw;
  const float* tw;
  const float* tb;
  float* aggr;              // [R, 2M + D + T]
  int64_t* new_lu;          // [R]
  long long R;
  int M, D, T, N, mean;
  int64_t* assoc;           // optional [N]: assoc[node] = (stamp << 32) | row  (tgn.py:193: self._assoc[n_id] = arange)
  long long stamp;
  float* h_out;             // optional [R, M]: h_out[row] = memory[nodes[row]] (the GRU's hidden-state operand: no gather launch)
};

__device__ __forceinline__ void tgn_message_cols(const AggrArgs& a, int lane, const float* mem_v, long long ev, long long lu_v,
                                                 float scale, float* out, bool accumulate) {
  const int M = a.M, D = a.D, T = a.T;
  const int other = a.log_other[ev];
  const float* mem_o = a.memory + (long long)other * M;
  const float* raw = a.log_raw + ev * D;
  const float dt = (float)(a.log_t[ev] - lu_v);
  for (int c = lane; c < 2 * M + D + T; c += kWave) {
    float v;
    if (c < M) v = mem_v[c];
    else if (c < 2 * M) v = mem_o[c - M];
    else if (c < 2 * M + D) v = raw[c - 2 * M];
    else v = cos_t2v(__fmaf_rn(dt, a.tw[c - 2 * M - D], a.tb[c - 2 * M - D]));
    out[c] = accumulate ? out[c] + v * scale : v * scale;
  }
}

__global__ __launch_bounds__(256) void tgn_aggregate_kernel(const AggrArgs a) {
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= a.R) return;
  const int lane = lane_id();
  const int W = 2 * a.M + a.D + a.T;
  float* out = a.aggr + row * W;
  int v = a.nodes[row];
  if (v < 0) v += a.N;
  if (a.assoc && lane == 0) a.assoc[v] = (a.stamp << 32) | row;
  const long long lu_v = a.last_update[v];
  const float* mem_v = a.memory + (long long)v * a.M;
  const long long lo0 = a.st_lo[0][v], lo1 = a.st_lo[1][v];
  const int c0 = a.st_cnt[0][v], c1 = a.st_cnt[1][v];
  const int total = c0 + c1;
  if (a.h_out)
    for (int c = lane; c < a.M; c += kWave) a.h_out[row * a.M + c] = mem_v[c];
  if (total == 0) {
    for (int c = lane; c < W; c += kWave) out[c] = 0.f;
    if (lane == 0) a.new_lu[row] = 0;
    return;
  }
  // walk order index i -> log position
  auto pos = [&](int i) -> long long { return i < c0 ? lo0 + i : lo1 + (i - c0); };
  // max int64 t (last_update) and, for LAST, argmax of float32(t) with the smallest walk index on ties
  long long tmax = -0x7fffffffffffffffLL;
  float fbest = -__builtin_inff();
  int ibest = 0x7fffffff;
  for (int i = lane; i < total; i += kWave) {
    const long long t = a.log_t[pos(i)];
    tmax = t > tmax ? t : tmax;
    const float f = (float)t;
    if (f > fbest) {  // strictly greater: earlier index wins within a lane (indices ascend)
      fbest = f;
      ibest = i;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const long long t2 = __shfl_xor(tmax, o);
    tmax = t2 > tmax ? t2 : tmax;
    const float f2 = __shfl_xor(fbest, o);
    const int i2 = __shfl_xor(ibest, o);
    if (f2 > fbest || (f2 == fbest && i2 < ibest)) {
      fbest = f2;
      ibest = i2;
    }
  }
  if (lane == 0) a.new_lu[row] = tmax;
  if (!a.mean) {
    tgn_message_cols(a, lane, mem_v, pos(ibest), lu_v, 1.f, out, false);
  } else {
    // sum in walk order, then divide (scatter-mean: sum / count)
    for (int i = 0; i < total; ++i) tgn_message_cols(a, lane, mem_v, pos(i), lu_v, 1.f, out, i > 0);
    const float inv = (float)total;
    for (int c = lane; c < W; c += kWave) out[c] = out[c] / inv;
  }
}

// GRUCell gates (torch.nn.GRUCell): gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh (both [R, 3M], r|z|n)
__global__ __launch_bounds__(256) void tgn_gru_gate_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h, int M, long long R,
                                                           float* __restrict__ out) {
  const long long total = R * M;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long r = e / M;
    const int c = (int)(e - r * M);
    const float* gir = gi + r * 3 * M;
    const float* ghr = gh + r * 3 * M;
    const float rg = 1.f / (1.f + expf(-(gir[c] + ghr[c])));
    const float zg = 1.f / (1.f + expf(-(gir[M + c] + ghr[M + c])));
    const float ng = tanhf(gir[2 * M + c] + rg * ghr[2 * M + c]);
    out[e] = (1.f - zg) * ng + zg * h[e];
  }
}

// memory[nodes[r]] = val[r], last_update[nodes[r]] = lu[r]  where flag[r] (null = all rows)
__global__ __launch_bounds__(256) void tgn_commit_kernel(const int32_t* __restrict__ nodes, const unsigned char* __restrict__ flag,
                                                         const float* __restrict__ val, const int64_t* __restrict__ lu, int M, int N,
                                                         long long R, float* __restrict__ memory, int64_t* __restrict__ last_update) {
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R || (flag && !flag[row])) return;
  int v = nodes[row];
  if (v < 0) v += N;
  for (int c = lane_id(); c < M; c += kWave) memory[(long long)v * M + c] = val[row * M + c];
  if (lane_id() == 0) last_update[v] = lu[row];
}

// ---------------------------------------------------------------------------
// GraphAttentionEmbedding (tgn.py:14-40) = Time2Vec edge encoding + TransformerConv.
// TransformerConv is third-party (torch_geometric 2.6.1); implemented from its published definition:
//   alpha_ij = softmax_i( q_i . (k_j + e_ij) / sqrt(C) ),  out_i = sum_j alpha_ij (v_j + e_ij)  per head,
//   heads concatenated, plus the root/skip projection (added by the caller's GEMM epilogue).
// ---------------------------------------------------------------------------
// edge_attr[e] = [cos(fma(float(last_update[src_local[e]] - t[e]), w, b)) | msg[e, :]]
// tgt / tgt_count (optional): the thread that writes an edge's first column also counts the edge for its target (the first pass of
// the counting grouping in tgmx_tconv_forward); a target outside [0, U) is clamped and flagged like tgmx_segment_sort does.
// rel_zero (optional): [U] zeroed for the fused scan + placement that follows.
__global__ __launch_bounds__(256) void tconv_edge_attr_kernel(const int64_t* __restrict__ lu_local, const int64_t* __restrict__ src,
                                                              const int64_t* __restrict__ t, const float* __restrict__ msg,
                                                              const float* __restrict__ tw, const float* __restrict__ tb, int T,
                                                              int D, long long E, float* __restrict__ out, const int64_t* __restrict__ tgt,
                                                              int32_t* __restrict__ tgt_count, long long U, int32_t* status,
                                                              int64_t* __restrict__ rel_zero) {
  const int W = T + D;
  const long long total = E * W;
  const long long step = (long long)gridDim.x * blockDim.x;
  if (rel_zero) {  // the running count per target of tconv_group_scan_place_kernel, one launch on
    for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < U; u += step) rel_zero[u] = 0;
  }
  for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += step) {
    const long long e = x / W;
    const int c = (int)(x - e * W);
    if (c == 0 && tgt_count) {
      long long i = tgt[e];
      if (i < 0 || i >= U) {
        atomicOr(status, TGMX_ST_EDGE_RANGE);
        i = i < 0 ? 0 : U - 1;
      }
      atomicAdd(&tgt_count[i], 1);
    }
    float v;
    if (c < T) v = cos_t2v(__fmaf_rn((float)(lu_local[src[e]] - t[e]), tw[c], tb[c]));
    else v = msg[e * D + (c - T)];
    out[x] = v;
  }
}

// The memory commit of a TGN step (tgn_commit_assoc_kernel below: 2n waves, a row copy each) as extra workgroups of the attention's launch:
// it reads the look-ahead rows the GRU gates wrote two launches earlier and writes `memory` / `last_update`, which the attention does not
// touch -- one launch less per batch on a host-bound step (tgmx_tgn_step).  n = 0: no rider.
struct CommitRider {
  const int32_t* src;
  const int32_t* dst;
  const int64_t* assoc;
  const float* val;
  const int64_t* lu;
  float* memory;
  int64_t* last_update;
  int32_t* status;
  long long n, stamp;
  int M, N;
};

struct TconvArgs {
  const float* q;      // [U, H*C]
  const float* k;      // [U, H*C]
  const float* v;      // [U, H*C]
  const float* eproj;  // [E, H*C] rows in ORIGINAL edge order
  const int64_t* order;   // [E] edge ids sorted by target
  const int64_t* src;     // [E] source (j) of every edge, original order
  const int64_t* seg_lo;  // [U] first position in `order` of target i's incoming edges
  const int64_t* seg_hi;  // [U]
  float* out;          // [U, H*C], pre-filled with the skip projection; attention output is ADDED
  long long U;
  int H, C;
  float scale;
  DropoutArgs drop;  // training: dropout on the attention coefficients after the softmax (PyG TransformerConv), element e * H + h
  // counting grouping (tgmx_tconv_forward): `order` holds every target's incoming edge ids in NO particular order (placed with
  // atomics); the workgroup sorts its segment by edge id first -- ascending edge id IS the stable order a sort by target gives --
  // in LDS, or for a segment longer than kTconvSegLds into order_big[lo .. hi)
  int unsorted;
  int64_t* order_big;
  int short_ok;  // floats per lane of the register-resident walks (4 or 2), 0 = the generic walk (tconv_short_ok)
  CommitRider commit;  // workgroups U .. U + ceil(2n / 4) of the launch (tgmx_tgn_step)
  int32_t* count_zero;  // [U] or NULL: the grouping's histogram, cleared here when the fused scan + placement left it standing
};

// memory[v] = val[row(v)], last_update[v] = lu[row(v)] for batch entry e of [src | dst]; one wave per entry
__device__ __forceinline__ void commit_entry(const CommitRider& c, long long e) {
  if (e >= 2 * c.n) return;
  int v = e < c.n ? c.src[e] : c.dst[e - c.n];
  if (v < 0) v += c.N;
  const long long as = c.assoc[v];
  if ((as >> 32) != c.stamp) {
    if (lane_id() == 0) atomicOr(c.status, 1);
    return;
  }
  const long long row = as & 0xffffffffll;
  for (int col = lane_id(); col < c.M; col += kWave) c.memory[(long long)v * c.M + col] = c.val[row * c.M + col];
  if (lane_id() == 0) c.last_update[v] = c.lu[row];
}

// The register-resident walks apply to H = 2 heads of C = V c columns, c <= 32, V = 4 or 2 floats per lane (cfg 3: C = 50 -> V = 2), rows at
// V-float-aligned addresses: returns V, or 0 for the generic walk
static int tconv_short_ok(const TconvArgs& a) {
  static const bool knob = !(getenv("TGMX_TCONV_SHORT") && atoi(getenv("TGMX_TCONV_SHORT")) == 0);  // A/B knob: 0 = the generic four-wave walk
  if (!knob || a.H != 2) return 0;
  const uintptr_t bits = (uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.eproj | (uintptr_t)a.out;
  if (a.C % 4 == 0 && a.C / 4 <= 32 && (bits & 15) == 0) return 4;
  if (a.C % 2 == 0 && a.C / 2 <= 32 && (bits & 7) == 0) return 2;
  return 0;
}

constexpr int kTconvSegLds = 1024;
constexpr int kTconvShort = 16;  // segments of up to this many edges take the one-wave, register-resident path below

// The register-resident walks of tconv_attend_kernel (round 5, last session).  A target's incoming edges on ONE wave in the TGAT attention
// row's layout: lane (h, c) = (lane >> 5, lane & 31) owns V consecutive columns (V = 4 or 2 floats: one 16- / 8-byte load; C / V <= 32) of
// head h of every row it touches (H = 2), the edges go through in blocks of four whose 12 row pieces (k, v of the source, the edge
// projection) are ALL requested before the previous block is scored, a score is a five-step butterfly inside the head's 32 lanes (DPP, no
// LDS) so every lane of a head holds the head's softmax state and weights its own columns -- no broadcast, no barrier.
//   * a target with <= kTconvShort edges (cfg 3: k = 10 sampled neighbors per hop -- nearly every target): wave 0 alone, ids ranked in registers;
//   * a longer segment (the review shape's hub items: hundreds of edges, whose walk SETS the launch's duration): the same walk on each of
//     the four waves over its quarter of the positions, the four states merged in LDS in wave order.
// What they replace (the generic walk below: one edge at a time, every edge an exposed round trip of 4-byte loads, a 64-lane shuffle
// reduction through the LDS crossbar per edge and head, three barriers and an LDS merge per head): 40.4 -> 13.9 us per cfg 3 batch,
// pipeline 200-201 -> 168-170 us (same box, alternating runs).  Same online-softmax recurrence over the edges in ascending edge id; the
// sums differ from the generic walk's at rounding level (another split of the columns over lanes): same tolerance against the restatement.
// V floats (2 or 4: one 8- / 16-byte load) of a row: the piece of a head's columns a lane owns
template <int V>
struct TcVec {
  float f[V];
};
template <int V>
__device__ __forceinline__ TcVec<V> tc_load(const float* __restrict__ p) {
  TcVec<V> r;
  if constexpr (V == 4) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    r.f[0] = x.x; r.f[1] = x.y; r.f[2] = x.z; r.f[3] = x.w;
  } else {
    const float2 x = *reinterpret_cast<const float2*>(p);
    r.f[0] = x.x; r.f[1] = x.y;
  }
  return r;
}
template <int V>
__device__ __forceinline__ void tc_store(float* __restrict__ p, const TcVec<V>& r) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r.f[0], r.f[1], r.f[2], r.f[3]);
  else *reinterpret_cast<float2*>(p) = make_float2(r.f[0], r.f[1]);
}
template <int V>
__device__ __forceinline__ TcVec<V> tc_zero() {
  TcVec<V> r;
#pragma unroll
  for (int u = 0; u < V; ++u) r.f[u] = 0.f;
  return r;
}

// the online-softmax state of one wave over the edges it has walked so far: every lane of a head holds the head's (m, l), a lane its columns' sums
template <int V>
struct TconvWalk {
  float m, l;
  TcVec<V> acc;
};

// Walk the n_here (<= 64) edges whose ids / sources lanes 0 .. n_here - 1 hold, in lane order, four at a time.  col = the lane's first column of
// the H C-wide row (its head's h C + V (lane & 31)), `on` = the lane owns columns at all.
template <int V>
__device__ __forceinline__ void tconv_walk_block(const TconvArgs& a, const TcVec<V> q, const bool on, const int col, const int h, const int my_e,
                                                 const int my_j32, const int n_here, TconvWalk<V>& st) {
  const long long HC = (long long)a.H * a.C;
  const float* __restrict__ kp = a.k + col;
  const float* __restrict__ vp = a.v + col;
  const float* __restrict__ ep = a.eproj + col;
  TcVec<V> kk[4], vv[4], ee[4], kn[4], vn[4], en[4];
  auto request = [&](int c0, TcVec<V> (&K)[4], TcVec<V> (&Vv)[4], TcVec<V> (&E)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int sl = c0 + s < n_here ? c0 + s : n_here - 1;  // (clamped: a position past the block re-reads its last edge, not used below)
      const long long e = __builtin_amdgcn_readlane(my_e, sl), j = __builtin_amdgcn_readlane(my_j32, sl);
      K[s] = on ? tc_load<V>(kp + j * HC) : tc_zero<V>();
      Vv[s] = on ? tc_load<V>(vp + j * HC) : tc_zero<V>();
      E[s] = on ? tc_load<V>(ep + e * HC) : tc_zero<V>();
    }
  };
  request(0, kk, vv, ee);
  float m = st.m, l = st.l;
  TcVec<V> acc = st.acc;
  for (int c0 = 0; c0 < n_here; c0 += 4) {
    const bool more = c0 + 4 < n_here;
    if (more) request(c0 + 4, kn, vn, en);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (c0 + s < n_here) {  // wave-uniform
        float p = q.f[0] * (kk[s].f[0] + ee[s].f[0]);
#pragma unroll
        for (int u = 1; u < V; ++u) p = __fmaf_rn(q.f[u], kk[s].f[u] + ee[s].f[u], p);
        p += lanes::lane_xor<16>(p);  // the head's 32 lanes (lanes past C / V carry zeros)
        p += lanes::lane_xor<8>(p);
        p += lanes::lane_xor<4>(p);
        p += lanes::lane_xor<2>(p);
        p += lanes::lane_xor<1>(p);
        const float sc = p * a.scale;
        const float mn = sc > m ? sc : m;
        const float corr = expf(m - mn), w = expf(sc - mn);
        const int e = __builtin_amdgcn_readlane(my_e, c0 + s);
        // the softmax normalises over every incoming edge; dropout then zeroes / rescales single coefficients
        const float wd = w * dropout_scale(a.drop, (unsigned long long)e * a.H + h);
#pragma unroll
        for (int u = 0; u < V; ++u) acc.f[u] = acc.f[u] * corr + wd * (vv[s].f[u] + ee[s].f[u]);
        l = l * corr + w;
        m = mn;
      }
    }
    if (more) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kk[s] = kn[s];
        vv[s] = vn[s];
        ee[s] = en[s];
      }
    }
  }
  st.m = m;
  st.l = l;
  st.acc = acc;
}

template <int V>
__device__ __forceinline__ void tconv_attend_short(const TconvArgs& a, const long long i, const long long lo, const int n, const int lane) {
  const int CV = a.C / V;
  const long long HC = (long long)a.H * a.C;
  const int h = lane >> 5, cl = lane & 31;
  const bool on = cl < CV;
  const int col = h * a.C + V * (on ? cl : 0);
  // the segment's edge ids, ascending (lane t holds the t-th), and their sources
  int my_e = 0;
  if (lane < n) my_e = (int)a.order[lo + lane];  // (edge ids of one batch: far below 2^31)
  if (a.unsorted && n > 1) {  // rank = number of smaller ids (ids are distinct); the id travels to the lane of its rank
    int rank = 0;
#pragma unroll
    for (int t = 0; t < kTconvShort; ++t) {
      const int o = __builtin_amdgcn_readlane(my_e, t);
      rank += (t < n && o < my_e) ? 1 : 0;
    }
    my_e = __builtin_amdgcn_ds_permute((lane < n ? rank : lane) << 2, my_e);
  }
  long long my_j = 0;
  if (lane < n) my_j = a.src[my_e];
  const TcVec<V> q = on ? tc_load<V>(a.q + i * HC + col) : tc_zero<V>();
  TconvWalk<V> st{-__builtin_inff(), 0.f, tc_zero<V>()};
  tconv_walk_block<V>(a, q, on, col, h, my_e, (int)my_j, n, st);
  if (on) {
    float* __restrict__ op = a.out + i * HC + col;
    TcVec<V> o = tc_load<V>(op);
#pragma unroll
    for (int u = 0; u < V; ++u) o.f[u] += st.acc.f[u] / st.l;
    tc_store<V>(op, o);
  }
}

// a LONG segment (a hub: hundreds of incoming edges at the review shape, whose walk sets the launch's duration) with the same block walk on
// each of the workgroup's four waves: wave w takes positions lo + w, lo + w + 4, ... 64 at a time (lane t fetches the id and the source of
// the t-th), the four states meet in LDS and are merged in wave order (deterministic).  ids(p) = the segment's p-th edge id, ascending.
template <int V, class Ids>
__device__ __forceinline__ void tconv_attend_long(const TconvArgs& a, const long long i, const long long lo, const long long hi, const int lane,
                                                  const int wave, Ids ids, float (*s_wm)[2], float (*s_wl)[2], float (*s_wacc)[kWave][4]) {
  constexpr int kWavesPerTarget = 4;
  const int CV = a.C / V;
  const long long HC = (long long)a.H * a.C;
  const int h = lane >> 5, cl = lane & 31;
  const bool on = cl < CV;
  const int col = h * a.C + V * (on ? cl : 0);
  const TcVec<V> q = on ? tc_load<V>(a.q + i * HC + col) : tc_zero<V>();
  TconvWalk<V> st{-__builtin_inff(), 0.f, tc_zero<V>()};
  for (long long p0 = lo + wave; p0 < hi; p0 += (long long)kWavesPerTarget * kWave) {
    const long long my_p = p0 + (long long)kWavesPerTarget * lane;
    int my_e = 0;
    long long my_j = 0;
    if (my_p < hi) {
      my_e = ids(my_p - lo);
      my_j = a.src[my_e];
    }
    const long long left = (hi - p0 + kWavesPerTarget - 1) / kWavesPerTarget;
    tconv_walk_block<V>(a, q, on, col, h, my_e, (int)my_j, left < kWave ? (int)left : kWave, st);
  }
  if (cl == 0) {
    s_wm[wave][h] = st.m;
    s_wl[wave][h] = st.l;
  }
#pragma unroll
  for (int u = 0; u < V; ++u) s_wacc[wave][lane][u] = st.acc.f[u];
  __syncthreads();
  if (wave == 0 && on) {
    float M = s_wm[0][h];
#pragma unroll
    for (int w2 = 1; w2 < kWavesPerTarget; ++w2) M = fmaxf(M, s_wm[w2][h]);
    float L = 0.f;
    TcVec<V> A = tc_zero<V>();
#pragma unroll
    for (int w2 = 0; w2 < kWavesPerTarget; ++w2) {
      const float f = s_wl[w2][h] > 0.f ? expf(s_wm[w2][h] - M) : 0.f;  // a wave without edges has m = -inf, l = 0
      L += s_wl[w2][h] * f;
#pragma unroll
      for (int u = 0; u < V; ++u) A.f[u] += s_wacc[w2][lane][u] * f;
    }
    float* __restrict__ op = a.out + i * HC + col;
    TcVec<V> o = tc_load<V>(op);
#pragma unroll
    for (int u = 0; u < V; ++u) o.f[u] += A.f[u] / L;
    tc_store<V>(op, o);
  }
}

// online softmax over a target's incoming edges, head by head
__global__ __launch_bounds__(256) void tconv_attend_kernel(const TconvArgs a) {
  // ONE WORKGROUP (4 waves) per target: wave w takes the target's incoming edges lo + w, lo + w + 4, ... and keeps an online
  // softmax partial (running max m, normaliser l, weighted sum acc per output column); the four partials meet in LDS and
  // are merged in wave order (deterministic).  With a wave per target the launch lasted as long as the busiest hub's
  // edge list walked serially (116 us at the review shape: items with hundreds of incoming edges).
  constexpr int kWavesPerTarget = 4;
  __shared__ float s_m[kWavesPerTarget], s_l[kWavesPerTarget], s_acc[kWavesPerTarget][kWave];
  const long long i = blockIdx.x;
  if (i >= a.U) {
    if (a.commit.n) commit_entry(a.commit, (i - a.U) * kWavesPerTarget + (threadIdx.x >> 6));
    return;
  }
  if (a.count_zero && threadIdx.x == 0) a.count_zero[i] = 0;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const long long lo = a.seg_lo[i], hi = a.seg_hi[i];
  if (hi <= lo) return;  // no incoming edge: only the skip term
  if (a.short_ok && hi - lo <= kTconvShort) {  // few edges: one wave, everything in registers (the other three waves are done)
    if (wave == 0) {
      if (a.short_ok == 4) tconv_attend_short<4>(a, i, lo, (int)(hi - lo), lane);
      else tconv_attend_short<2>(a, i, lo, (int)(hi - lo), lane);
    }
    return;
  }
  const int HC = a.H * a.C;
  __shared__ int s_raw[kTconvSegLds], s_sorted[kTconvSegLds];  // (edge ids of one batch: far below 2^31)
  const int64_t* ord = a.order + lo;  // position p of the segment reads ord[p - lo] / s_sorted[p - lo] (no __restrict__: the hub path writes what it reads)
  bool in_lds = false;
  if (a.unsorted && hi - lo > 1) {
    const long long n = hi - lo;
    if (n <= kTconvSegLds) {  // rank = number of smaller ids (ids are distinct): O(n^2 / 256) LDS reads, n is 1-3 for most targets
      for (int p = threadIdx.x; p < n; p += blockDim.x) s_raw[p] = (int)a.order[lo + p];
      __syncthreads();
      for (int p = threadIdx.x; p < n; p += blockDim.x) {
        const int id = s_raw[p];
        int rank = 0;
        for (int q = 0; q < n; ++q) rank += s_raw[q] < id;
        s_sorted[rank] = id;
      }
      __syncthreads();
      in_lds = true;
    } else {
      // a hub with more than kTconvSegLds incoming edges: runs of kTconvSegLds ids ranked in LDS like above, then merged pairwise --
      // an id's place in the merged run is its place in its own run + the number of smaller ids in the sibling run (binary search;
      // ids are distinct) -- ping-pong between this target's slice of `order` (this workgroup's alone) and of order_big:
      // O(n log n) instead of the O(n^2) ranking a 20 k-edge hub would spend hundreds of milliseconds in.
      int64_t* src_buf = const_cast<int64_t*>(a.order) + lo;
      int64_t* dst_buf = a.order_big + lo;
      for (long long c0 = 0; c0 < n; c0 += kTconvSegLds) {
        const int m = (int)((n - c0) < kTconvSegLds ? (n - c0) : kTconvSegLds);
        for (int p = threadIdx.x; p < m; p += blockDim.x) s_raw[p] = (int)src_buf[c0 + p];
        __syncthreads();
        for (int p = threadIdx.x; p < m; p += blockDim.x) {
          const int id = s_raw[p];
          int rank = 0;
          for (int q = 0; q < m; ++q) rank += s_raw[q] < id;
          s_sorted[rank] = id;
        }
        __syncthreads();
        for (int p = threadIdx.x; p < m; p += blockDim.x) dst_buf[c0 + p] = s_sorted[p];
        __syncthreads();
      }
      int64_t* t = src_buf; src_buf = dst_buf; dst_buf = t;  // src_buf: sorted runs of kTconvSegLds
      for (long long run = kTconvSegLds; run < n; run *= 2) {
        __threadfence_block();
        __syncthreads();
        for (long long p = threadIdx.x; p < n; p += blockDim.x) {
          const long long base = p / (2 * run) * (2 * run);
          const bool right = p - base >= run;
          const long long own_lo = right ? base + run : base;
          long long sl = right ? base : base + run, sh = sl + run;
          if (sl > n) sl = n;
          if (sh > n) sh = n;
          const int64_t id = src_buf[p];
          long long b = sl, e = sh;  // first position of the sibling run whose id is not smaller
          while (b < e) {
            const long long mid = (b + e) >> 1;
            if (src_buf[mid] < id) b = mid + 1; else e = mid;
          }
          dst_buf[base + (p - own_lo) + (b - sl)] = id;
        }
        t = src_buf; src_buf = dst_buf; dst_buf = t;
      }
      __threadfence_block();
      __syncthreads();
      ord = src_buf;
    }
  }
  if (a.short_ok) {
    __shared__ float s_wm[kWavesPerTarget][2], s_wl[kWavesPerTarget][2];
    __shared__ float s_wacc[kWavesPerTarget][kWave][4];
    auto ids = [&](long long p) -> int { return in_lds ? s_sorted[p] : (int)ord[p]; };
    if (a.short_ok == 4) tconv_attend_long<4>(a, i, lo, hi, lane, wave, ids, s_wm, s_wl, s_wacc);
    else tconv_attend_long<2>(a, i, lo, hi, lane, wave, ids, s_wm, s_wl, s_wacc);
    return;
  }
  for (int h = 0; h < a.H; ++h) {
    for (int c0 = 0; c0 < a.C; c0 += kWave) {  // output columns of this head handled by this lane
      // (C <= 64 in practice: one pass; for larger C the scores are recomputed per column chunk)
      const int c = c0 + lane;
      float m = -__builtin_inff(), l = 0.f, acc = 0.f;
      // this wave's edges 64 at a time: lane t fetches (edge id, source) of position p0 + 4 t, handed out by shuffles
      for (long long p0 = lo + wave; p0 < hi; p0 += (long long)kWavesPerTarget * kWave) {
        const long long my_p = p0 + (long long)kWavesPerTarget * lane;
        long long my_e = 0, my_j = 0;
        if (my_p < hi) {
          my_e = in_lds ? s_sorted[my_p - lo] : ord[my_p - lo];
          my_j = a.src[my_e];
        }
        const long long left = (hi - p0 + kWavesPerTarget - 1) / kWavesPerTarget;
        const int n_here = left < kWave ? (int)left : kWave;
        for (int t = 0; t < n_here; ++t) {
        const long long e = __shfl(my_e, t), j = __shfl(my_j, t);
        float part = 0.f;
        for (int cc = lane; cc < a.C; cc += kWave) {
          const int col = h * a.C + cc;
          part = __fmaf_rn(a.q[i * HC + col], a.k[j * HC + col] + a.eproj[e * HC + col], part);
        }
        const float val = c < a.C ? a.v[j * HC + h * a.C + c] + a.eproj[e * HC + h * a.C + c] : 0.f;
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        const float s = part * a.scale;
        const float mn = s > m ? s : m;
        const float corr = expf(m - mn), w = expf(s - mn);
        // the softmax normalises over every incoming edge; dropout then zeroes / rescales single coefficients
        acc = acc * corr + w * dropout_scale(a.drop, (unsigned long long)e * a.H + h) * val;
        l = l * corr + w;
        m = mn;
        }
      }
      __syncthreads();  // the previous (head, column chunk)'s partials have been consumed
      if (lane == 0) {
        s_m[wave] = m;
        s_l[wave] = l;
      }
      s_acc[wave][lane] = acc;
      __syncthreads();
      if (wave == 0) {
        float M = s_m[0];
#pragma unroll
        for (int w2 = 1; w2 < kWavesPerTarget; ++w2) M = fmaxf(M, s_m[w2]);
        float L = 0.f, A = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < kWavesPerTarget; ++w2) {
          const float f = s_l[w2] > 0.f ? expf(s_m[w2] - M) : 0.f;  // a wave without edges has m = -inf, l = 0
          L += s_l[w2] * f;
          A += s_acc[w2][lane] * f;
        }
        if (c < a.C) a.out[i * HC + h * a.C + c] += A / L;
      }
    }
  }
}

// Group a small id array (n <= 1024) in ONE launch: stable sort by id (the entry index rides in the key's low bits),
// the permutation, every position's run bounds and a first-of-run flag -- what the host otherwise assembles from
// torch.sort + 2 x searchsorted (message store, tgn.py:218-229) or sort + compare (unique commit rows, tgn.py:165-177).
__global__ __launch_bounds__(1024) void group_ids_kernel(const int32_t* __restrict__ ids, int n, int32_t* __restrict__ sorted,
                                                         int64_t* __restrict__ perm, int64_t* __restrict__ run_lo,
                                                         int64_t* __restrict__ run_hi, unsigned char* __restrict__ first) {
  __shared__ long long s_key[1024];
  __shared__ int s_pay[1024];
  __shared__ int s_id[1024];
  __shared__ int s_start[1024];
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, P = blockDim.x;
  long long key = 0x7fffffffffffffffLL;
  int pay = tid;
  if (tid < n) key = (((long long)ids[tid] + (1ll << 31)) << kPackBits) | (long long)tid;  // ids may be negative (pads)
  bitonic_sort_one<true>(key, pay, s_key, s_pay, tid, P);
  const int id = tid < n ? (int)((key >> kPackBits) - (1ll << 31)) : 0x7fffffff;
  s_id[tid] = id;
  __syncthreads();
  // run start = last position <= tid that opens a run (max-scan thread -> wave -> block)
  const bool opens = tid == 0 || s_id[tid - 1] != id;
  int incl = opens ? tid : 0;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl = o > incl ? o : incl;
  }
  if (lane == kWave - 1) wave_tot[wave] = incl;
  __syncthreads();
  for (int w = 0; w < wave; ++w) incl = wave_tot[w] > incl ? wave_tot[w] : incl;
  s_start[tid] = incl;
  __syncthreads();
  if (tid >= n) return;
  // run end: first later position that opens a run (walk is short: runs are per-node event counts of one batch)
  int hi = tid + 1;
  while (hi < n && s_start[hi] == incl) ++hi;
  if (sorted) sorted[tid] = id;
  if (perm) perm[tid] = pay;
  if (run_lo) run_lo[tid] = incl;
  if (run_hi) run_hi[tid] = hi;
  if (first) first[tid] = opens ? 1 : 0;
}


// ---- update_state in train mode with the rows of the preceding forward (tgn.py:165-177) ------------------------------
// The reference commits `_get_updated_memory(unique(src, dst))` -- the same rows its forward just produced for those
// nodes (same state, same arithmetic).  With `assoc` filled by that forward's aggregation, the commit is a row copy:
// memory[v] = val[row(v)], last_update[v] = lu[row(v)] for every batch entry (duplicates write the same values, so no
// unique / first-occurrence pass).  An entry whose node was not part of that forward (stale stamp) raises *status.
__global__ __launch_bounds__(256) void tgn_commit_assoc_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, long long n,
                                                               const int64_t* __restrict__ assoc, long long stamp,
                                                               const float* __restrict__ val, const int64_t* __restrict__ lu, int M, int N,
                                                               float* __restrict__ memory, int64_t* __restrict__ last_update,
                                                               int32_t* __restrict__ status) {
  const CommitRider c{src, dst, assoc, val, lu, memory, last_update, status, n, stamp, M, N};
  commit_entry(c, (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
}

// ---- message store of one batch, BOTH roles, one launch (tgn.py:218-229 called twice, :173,176) -----------------------
// Workgroup `role` (0: events keyed by src, other = dst; 1: keyed by dst, other = src) sorts its n <= 1024 keys stably
// (the entry index rides in the key), finds every key's run, and writes the log rows [base + role * n, +n) in sorted order
// plus the node's (lo, cnt) window -- what group_ids_kernel + tgn_store_kernel did as two launches per role.
struct StoreBatchArgs {
  const int32_t* src;
  const int32_t* dst;
  const int64_t* t;
  const float* raw;  // [n, D]
  int32_t* log_other;
  int64_t* log_t;
  float* log_raw;
  int64_t* st_lo[2];
  int32_t* st_cnt[2];
  long long base;
  int n, D;
};

struct StoreLds {
  long long key[1024];
  int pay[1024], id[1024], start[1024], perm[1024], wave_tot[16];
};

// one role's workgroup of P threads (a power of two >= n); the result does not depend on P (keys are unique: the entry index rides in them)
__device__ __forceinline__ void store_batch_body(const StoreBatchArgs& a, const int role, StoreLds& L, const int P) {
  long long* s_key = L.key;
  int *s_pay = L.pay, *s_id = L.id, *s_start = L.start, *s_perm = L.perm, *wave_tot = L.wave_tot;
  const int32_t* ids = role == 0 ? a.src : a.dst;
  const int32_t* oth = role == 0 ? a.dst : a.src;
  const int n = a.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long key = 0x7fffffffffffffffLL;
  int pay = tid;
  if (tid < n) key = (((long long)ids[tid] + (1ll << 31)) << kPackBits) | (long long)tid;
  bitonic_sort_one<true>(key, pay, s_key, s_pay, tid, P);
  const int id = tid < n ? (int)((key >> kPackBits) - (1ll << 31)) : 0x7fffffff;
  s_id[tid] = id;
  s_perm[tid] = pay;
  __syncthreads();
  const bool opens = tid == 0 || s_id[tid - 1] != id;
  int incl = opens ? tid : 0;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl = o > incl ? o : incl;
  }
  if (lane == kWave - 1) wave_tot[wave] = incl;
  __syncthreads();
  for (int w = 0; w < wave; ++w) incl = wave_tot[w] > incl ? wave_tot[w] : incl;
  s_start[tid] = incl;
  __syncthreads();
  const long long base = a.base + (long long)role * n;
  if (tid < n) {
    // one writer per node: the LAST entry of the run knows both ends (its own position and the run's start) -- the first entry would have
    // to walk to the end of the run (a hub's run of 50+ entries, read one by one)
    if (tid == n - 1 || s_id[tid + 1] != id) {
      a.st_lo[role][id] = base + incl;
      a.st_cnt[role][id] = tid + 1 - incl;
    }
    a.log_other[base + tid] = oth[pay];
    a.log_t[base + tid] = a.t[pay];
  }
  // raw rows in sorted order: four independent elements per thread and trip, all loaded before the first is stored (the source and
  // the log may alias as far as the compiler knows: a load behind a store waits for it -- one round trip per element otherwise)
  const float* __restrict__ raw = a.raw;
  float* __restrict__ log_raw = a.log_raw;
  const int total = n * a.D;
  for (int x0 = 0; x0 < total; x0 += 4 * P) {
    float v[4];
    int xs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int x = x0 + u * P + tid;
      xs[u] = x;
      const int xc = x < total ? x : 0;
      const int p = xc / a.D, c = xc - p * a.D;
      v[u] = raw[(long long)s_perm[p] * a.D + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (xs[u] < total) log_raw[base * a.D + xs[u]] = v[u];
  }
}

__global__ __launch_bounds__(1024) void tgn_store_batch_kernel(const StoreBatchArgs a) {
  __shared__ StoreLds L;
  store_batch_body(a, blockIdx.x, L, blockDim.x);
}

// ---- message-log compaction (TGNMemory's append-only log, nn/tgn.py _compact_into) ---------------------------------------------
// The live windows -- per role and node (lo, cnt) -- move, in (role, node) order, to the front of fresh log tensors.  Host side this was
// ~25 torch ops over [num_nodes]-sized temporaries, 1.1 ms per compaction at the review shape (every ~75 batches: 15 us of every batch);
// here: one scan of the 2 N counts and one move launch.
struct CompactCount {
  const int32_t* cnt_s;
  const int32_t* cnt_d;
  int N;
  __device__ long long operator()(int i) const { return (long long)(i < N ? cnt_s[i] : cnt_d[i - N]); }
};

__global__ __launch_bounds__(256) void tgn_compact_move_kernel(int64_t* __restrict__ lo_s, const int32_t* __restrict__ cnt_s, int64_t* __restrict__ lo_d,
                                                               const int32_t* __restrict__ cnt_d, int N, const long long* __restrict__ pos,
                                                               const int32_t* __restrict__ old_other, const int64_t* __restrict__ old_t,
                                                               const float* __restrict__ old_raw, int D, int32_t* __restrict__ new_other,
                                                               int64_t* __restrict__ new_t, float* __restrict__ new_raw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (role, node): role 0's nodes first
  if (i >= 2 * N) return;
  const int role = i >= N, v = role ? i - N : i;
  int64_t* lo = role ? lo_d : lo_s;
  const int c = role ? cnt_d[v] : cnt_s[v];
  if (c == 0) {
    lo[v] = 0;
    return;
  }
  const long long from = lo[v], to = pos[i];
  for (int j = 0; j < c; ++j) {
    new_other[to + j] = old_other[from + j];
    new_t[to + j] = old_t[from + j];
    for (int d = 0; d < D; ++d) new_raw[(to + j) * D + d] = old_raw[(from + j) * D + d];
  }
  lo[v] = to;
}

// ---- counting grouping of a batch's edges by target (tgmx_tconv_forward) -----------------------------------------------------
// count[U] (zero on entry: the histogram pass rode tconv_edge_attr_kernel) -> seg_lo / seg_hi / cursor = exclusive prefix sums, and
// count is zeroed again for the next call.  One workgroup, 16 consecutive counts per thread and pass (16 K targets per pass).
__global__ __launch_bounds__(1024) void tconv_group_scan_kernel(int32_t* __restrict__ count, long long U, int64_t* __restrict__ seg_lo,
                                                                int64_t* __restrict__ seg_hi, int64_t* __restrict__ cursor) {
  constexpr int kPer = 8;  // (8 K targets per pass: a review-shaped batch has ~8 000 unique nodes -- every thread busy, one pass)
  __shared__ long long wave_tot[16];
  __shared__ long long carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  const bool vec = ((reinterpret_cast<uintptr_t>(count) | reinterpret_cast<uintptr_t>(seg_lo) | reinterpret_cast<uintptr_t>(seg_hi) |
                     reinterpret_cast<uintptr_t>(cursor)) & 15) == 0;
  for (long long u0 = 0; u0 < U; u0 += 1024 * kPer) {
    const long long ub = u0 + (long long)tid * kPer;
    const bool full = vec && ub + kPer <= U;
    int c[kPer];
    if (full) {
      const int4 a = reinterpret_cast<const int4*>(count + ub)[0], b = reinterpret_cast<const int4*>(count + ub)[1];
      c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
    } else {
#pragma unroll
      for (int q = 0; q < kPer; ++q) c[q] = ub + q < U ? count[ub + q] : 0;
    }
    long long v = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) v += c[q];
    long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const long long o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    long long before = carry_s;
    for (int q = 0; q < wave; ++q) before += wave_tot[q];
    long long lo[kPer], hi[kPer];
    long long pos = before + incl - v;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      lo[q] = pos;
      pos += c[q];
      hi[q] = pos;
    }
    if (full) {
#pragma unroll
      for (int q = 0; q < kPer; q += 2) {
        const longlong2 l2{lo[q], lo[q + 1]}, h2{hi[q], hi[q + 1]};
        *reinterpret_cast<longlong2*>(seg_lo + ub + q) = l2;
        *reinterpret_cast<longlong2*>(cursor + ub + q) = l2;
        *reinterpret_cast<longlong2*>(seg_hi + ub + q) = h2;
      }
      if (v) {
        reinterpret_cast<int4*>(count + ub)[0] = int4{0, 0, 0, 0};
        reinterpret_cast<int4*>(count + ub)[1] = int4{0, 0, 0, 0};
      }
    } else {
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        if (ub + q < U) {
          seg_lo[ub + q] = lo[q];
          cursor[ub + q] = lo[q];
          seg_hi[ub + q] = hi[q];
          if (c[q]) count[ub + q] = 0;
        }
      }
    }
    __syncthreads();
    if (tid == 1023) carry_s = before + incl;
    __syncthreads();
  }
}

// order[cursor[tgt[e]]++] = e: every target's incoming edge ids land in its segment, in the order the atomics resolve
// (tconv_attend_kernel sorts a segment before it walks it)
__global__ __launch_bounds__(256) void tconv_group_place_kernel(const int64_t* __restrict__ tgt, long long E, long long U,
                                                                int64_t* __restrict__ cursor, int64_t* __restrict__ order) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  long long i = tgt[e];
  i = i < 0 ? 0 : (i >= U ? U - 1 : i);
  const unsigned long long p = atomicAdd(reinterpret_cast<unsigned long long*>(&cursor[i]), 1ull);
  order[p] = e;
}

// The two launches above as ONE, for U <= kGroupFusedMaxU targets: every workgroup of the placement scans the U counts itself (26 KB from L2
// at the review shape: cheaper than a launch boundary on a host-bound step) into LDS, then places its 1024 edges at seg_lo[target] + a running
// count per target.  `rel` [U] must be ZERO on entry (tconv_edge_attr_kernel zeroes it beside the histogram); `count` is read by every
// workgroup for the whole launch, so it is zeroed one launch later (tconv_attend_kernel: the workgroup of target i clears count[i]).
// Workgroup 0 also writes seg_lo / seg_hi for the attention.  store.n > 0: two more workgroups store the batch's messages (see below).
constexpr int kGroupFusedMaxU = 16000;  // (64 000 B of the 64 KB a workgroup may declare)
constexpr int kGroupFusedThreads = 1024;
__global__ __launch_bounds__(kGroupFusedThreads) void tconv_group_scan_place_kernel(const int64_t* __restrict__ tgt, long long E, int U,
                                                                                    const int32_t* __restrict__ count, int64_t* __restrict__ rel,
                                                                                    int64_t* __restrict__ seg_lo, int64_t* __restrict__ seg_hi,
                                                                                    int64_t* __restrict__ order, const StoreBatchArgs store) {
  constexpr int P = kGroupFusedThreads;
  static_assert(sizeof(StoreLds) <= kGroupFusedMaxU * sizeof(int), "the store rider's LDS lives in the scan's buffer");
  __shared__ __align__(16) int s_lo[kGroupFusedMaxU];
  __shared__ int wave_tot[P / 64];
  // tgmx_tgn_step: the batch's message store (tgn_store_batch_kernel: one workgroup per role) as the launch's LAST two workgroups -- it
  // shares nothing with the grouping but the stream position (behind the aggregation, in front of the join)
  const unsigned group_blocks = (unsigned)((E + P - 1) / P);
  if (blockIdx.x >= group_blocks) {
    store_batch_body(store, (int)(blockIdx.x - group_blocks), *reinterpret_cast<StoreLds*>(s_lo), P);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the counts into LDS with coalesced, independent loads (a thread walking its own chunk of global memory pays one L2 latency per count);
  // the target of this thread's edge is requested alongside
  const long long e = (long long)blockIdx.x * P + tid;
  long long i = e < E ? tgt[e] : 0;
#pragma unroll 16
  for (int u = tid; u < U; u += P) s_lo[u] = count[u];
  __syncthreads();
  const int per = (U + P - 1) / P;
  const int u0 = tid * per, u1 = u0 + per < U ? u0 + per : U;
  int v = 0;
  for (int u = u0; u < u1; ++u) v += s_lo[u];
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int pos = incl - v, total = 0;
  for (int q = 0; q < P / 64; ++q) {
    if (q < wave) pos += wave_tot[q];
    total += wave_tot[q];
  }
  for (int u = u0; u < u1; ++u) {  // (in place: the chunk's counts become its exclusive prefix sums)
    const int c = s_lo[u];
    s_lo[u] = pos;
    pos += c;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int u = tid; u < U; u += P) {
      seg_lo[u] = s_lo[u];
      seg_hi[u] = u + 1 < U ? s_lo[u + 1] : total;
    }
  }
  if (e >= E) return;
  i = i < 0 ? 0 : (i >= U ? U - 1 : i);
  const unsigned long long r = atomicAdd(reinterpret_cast<unsigned long long*>(&rel[i]), 1ull);
  order[s_lo[i] + (long long)r] = e;
}

// ---- the sampled edge list of one hop, as the reference's TGN loop builds it (examples/linkproppred/tgn.py:80-92) ------
//   mask = nbr != -1;  edge_index = [global_to_local(seed.repeat_interleave(k)[mask]); global_to_local(nbr[mask])]
//   edge_t = nbr_t[mask];  edge_x = nbr_x[mask]     -- order = slot order, local ids = position in the sorted unique ids
// Two launches: per-row valid counts + exclusive scan (one workgroup), then a wave per seed row writes its valid slots.
__global__ __launch_bounds__(1024) void edge_list_scan_kernel(const int32_t* __restrict__ nbr, long long S, int k, int64_t* __restrict__ row_off,
                                                              int64_t* __restrict__ count) {
  edge_list_scan_body(nbr, S, k, row_off, count);
}

struct EdgeListArgs {
  const int32_t* seed;   // [S]
  const int32_t* nbr;    // [S, k]
  const int64_t* nbr_t;  // [S, k]
  const float* nbr_x;    // [S, k, D]   (NULL with nbr_eid / table)
  const int32_t* nbr_eid;  // [S, k] edge id behind every slot (-1 = pad): the feature row is table[eid]   (edge features by id)
  const float* table;    // [E_store, D] the resident store's edge features
  const int32_t* uniq;   // [U] sorted unique ids
  const int64_t* uniq_count;  // device-side U (overrides the host value when set)
  const int64_t* row_off;
  int64_t* ei;           // [2, cap]
  int64_t* et;           // [cap]
  float* ex;             // [cap, D]
  long long S, U, cap;
  int k, D;
};

__device__ __forceinline__ long long local_id(const int32_t* __restrict__ uniq, long long U, int v) {  // searchsorted(left)
  long long lo = 0, hi = U;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (uniq[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void edge_list_write_kernel(const EdgeListArgs a) {
  const long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= a.S) return;
  const int lane = lane_id();
  const int k = a.k;
  const int id = lane < k ? a.nbr[r * k + lane] : -1;
  const int eid = (a.nbr_eid && lane < k) ? a.nbr_eid[r * k + lane] : -1;
  const bool ok = id != -1;
  const unsigned long long m = __ballot(ok);
  if (!m) return;
  const long long pos = a.row_off[r] + __popcll(m & ((1ull << lane) - 1ull));
  const long long U = a.uniq_count ? *a.uniq_count : a.U;
  const long long seed_local = local_id(a.uniq, U, a.seed[r]);
  if (ok) {
    a.ei[pos] = seed_local;
    a.ei[a.cap + pos] = local_id(a.uniq, U, id);
    a.et[pos] = a.nbr_t[r * k + lane];
  }
  // feature rows of the valid slots, in slot order
  unsigned long long rest = m;
  long long out = a.row_off[r];
  while (rest) {
    const int s = __ffsll((long long)rest) - 1;
    rest &= rest - 1;
    if (a.nbr_eid) {
      const int e = __shfl(eid, s);  // (a valid slot of a by-id sampler carries its edge's store id)
      const float* __restrict__ x = a.table + (long long)(e < 0 ? 0 : e) * a.D;
      for (int c = lane; c < a.D; c += kWave) a.ex[out * a.D + c] = e < 0 ? 0.f : x[c];
    } else {
      const float* __restrict__ x = a.nbr_x + (r * k + s) * (long long)a.D;
      for (int c = lane; c < a.D; c += kWave) a.ex[out * a.D + c] = x[c];
    }
    ++out;
  }
}

}  // namespace tgmx

using namespace tgmx;

extern "C" int tgmx_tgn_store(const int64_t* perm, const int32_t* node_sorted, const int64_t* left, const int64_t* right,
                              const int32_t* other, const int64_t* t, const float* raw, int32_t D, int64_t n, int64_t base,
                              int32_t* log_other, int64_t* log_t, float* log_raw, int64_t* st_lo, int32_t* st_cnt,
                              tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0 && D >= 0 && base >= 0, "tgn_store: bad sizes");
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(perm && node_sorted && left && right && other && t && (D == 0 || (raw && log_raw)) && log_other && log_t && st_lo && st_cnt,
               "tgn_store: null pointer");
  StoreArgs a{perm, node_sorted, left, right, other, t, raw, log_other, log_t, log_raw, st_lo, st_cnt, n, base, D};
  hipLaunchKernelGGL(tgn_store_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  TGMX_CHECK_LAUNCH("tgn_store");
  return TGMX_OK;
}

static int tgn_aggregate_impl(const int32_t* nodes, int64_t R, const float* memory, const int64_t* last_update, int32_t M,
                              int32_t num_nodes, const int64_t* st_lo_s, const int32_t* st_cnt_s, const int64_t* st_lo_d,
                              const int32_t* st_cnt_d, const int32_t* log_other, const int64_t* log_t, const float* log_raw,
                              int32_t D, const float* tw, const float* tb, int32_t T, int32_t mean, float* aggr, int64_t* new_lu,
                              int64_t* assoc, int64_t stamp, float* h_out, tgmx_stream_t stream) {
  TGMX_REQUIRE(R >= 0 && M > 0 && D >= 0 && T > 0 && num_nodes > 0, "tgn_aggregate: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(nodes && memory && last_update && st_lo_s && st_cnt_s && st_lo_d && st_cnt_d && tw && tb && aggr && new_lu,
               "tgn_aggregate: null pointer");
  AggrArgs a{};
  a.nodes = nodes; a.memory = memory; a.last_update = last_update;
  a.st_lo[0] = st_lo_s; a.st_lo[1] = st_lo_d; a.st_cnt[0] = st_cnt_s; a.st_cnt[1] = st_cnt_d;
  a.log_other = log_other; a.log_t = log_t; a.log_raw = log_raw; a.tw = tw; a.tb = tb; a.aggr = aggr; a.new_lu = new_lu;
  a.R = R; a.M = M; a.D = D; a.T = T; a.N = num_nodes; a.mean = mean;
  a.assoc = assoc; a.stamp = stamp; a.h_out = h_out;
  hipLaunchKernelGGL(tgn_aggregate_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  TGMX_CHECK_LAUNCH("tgn_aggregate");
  return TGMX_OK;
}

extern "C" int tgmx_tgn_aggregate(const int32_t* nodes, int64_t R, const float* memory, const int64_t* last_update, int32_t M,
                                  int32_t num_nodes, const int64_t* st_lo_s, const int32_t* st_cnt_s, const int64_t* st_lo_d,
                                  const int32_t* st_cnt_d, const int32_t* log_other, const int64_t* log_t, const float* log_raw,
                                  int32_t D, const float* tw, const float* tb, int32_t T, int32_t mean, float* aggr, int64_t* new_lu,
                                  int64_t* assoc, int64_t stamp, tgmx_stream_t stream) {
  return tgn_aggregate_impl(nodes, R, memory, last_update, M, num_nodes, st_lo_s, st_cnt_s, st_lo_d, st_cnt_d, log_other, log_t, log_raw, D, tw, tb, T,
                            mean, aggr, new_lu, assoc, stamp, nullptr, stream);
}

extern "C" int tgmx_tgn_gru_gate(const float* gi, const float* gh, const float* h, int32_t M, int64_t R, float* out,
                                 tgmx_stream_t stream) {
  TGMX_REQUIRE(M > 0 && R >= 0, "tgn_gru_gate: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(gi && gh && h && out, "tgn_gru_gate: null pointer");
  long long blocks = (R * M + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tgn_gru_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gi, gh, h, M, (long long)R, out);
  TGMX_CHECK_LAUNCH("tgn_gru_gate");
  return TGMX_OK;
}

extern "C" int tgmx_tgn_commit(const int32_t* nodes, const uint8_t* flag, const float* val, const int64_t* lu, int32_t M,
                               int32_t num_nodes, int64_t R, float* memory, int64_t* last_update, tgmx_stream_t stream) {
  TGMX_REQUIRE(M > 0 && R >= 0 && num_nodes > 0, "tgn_commit: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(nodes && val && lu && memory && last_update, "tgn_commit: null pointer");
  hipLaunchKernelGGL(tgn_commit_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, nodes, flag, val, lu, M,
                     num_nodes, (long long)R, memory, last_update);
  TGMX_CHECK_LAUNCH("tgn_commit");
  return TGMX_OK;
}

extern "C" int tgmx_tconv_edge_attr(const int64_t* last_update_local, const int64_t* src, const int64_t* t, const float* msg,
                                    const float* tw, const float* tb, int32_t T, int32_t D, int64_t E, float* out,
                                    tgmx_stream_t stream) {
  TGMX_REQUIRE(T > 0 && D >= 0 && E >= 0, "tconv_edge_attr: bad sizes");
  if (E == 0) return TGMX_OK;
  TGMX_REQUIRE(last_update_local && src && t && (D == 0 || msg) && tw && tb && out, "tconv_edge_attr: null pointer");
  long long blocks = (E * (T + D) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tconv_edge_attr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, last_update_local, src, t,
                     msg, tw, tb, T, D, (long long)E, out, (const int64_t*)nullptr, (int32_t*)nullptr, 0ll, (int32_t*)nullptr, (int64_t*)nullptr);
  TGMX_CHECK_LAUNCH("tconv_edge_attr");
  return TGMX_OK;
}

extern "C" int tgmx_tconv_attend(const float* q, const float* k, const float* v, const float* eproj, const int64_t* order,
                                 const int64_t* src, const int64_t* seg_lo, const int64_t* seg_hi, int64_t U, int32_t H, int32_t C,
                                 float scale, float* out, const tgmx_dropout_t* drop, tgmx_stream_t stream) {
  TGMX_REQUIRE(U >= 0 && H > 0 && C > 0, "tconv_attend: bad sizes");
  if (U == 0) return TGMX_OK;
  TGMX_REQUIRE(q && k && v && eproj && order && src && seg_lo && seg_hi && out, "tconv_attend: null pointer");
  TconvArgs a{q, k, v, eproj, order, src, seg_lo, seg_hi, out, U, H, C, scale};
  a.drop = make_dropout(drop);
  a.unsorted = 0;
  a.order_big = nullptr;
  a.short_ok = tconv_short_ok(a);
  hipLaunchKernelGGL(tconv_attend_kernel, dim3((unsigned)U), dim3(256), 0, (hipStream_t)stream, a);
  TGMX_CHECK_LAUNCH("tconv_attend");
  return TGMX_OK;
}

extern "C" int tgmx_group_ids(const int32_t* ids, int32_t n, int32_t* sorted, int64_t* perm, int64_t* run_lo, int64_t* run_hi,
                              uint8_t* first, tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0 && n <= 1024, "group_ids: n=%d (at most 1024 ids per call)", n);
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(ids, "group_ids: null pointer");
  int P = 64;
  while (P < n) P <<= 1;
  hipLaunchKernelGGL(group_ids_kernel, dim3(1), dim3(P), 0, (hipStream_t)stream, ids, n, sorted, perm, run_lo, run_hi, first);
  TGMX_CHECK_LAUNCH("group_ids");
  return TGMX_OK;
}

// ---- the same grouping for any n (a 4096-edge batch groups 8192 endpoint ids): one stable rocPRIM radix sort + a finishing launch ----
__global__ __launch_bounds__(256) void group_ids_finish_kernel(const int32_t* __restrict__ sorted, const unsigned* __restrict__ perm32, long long n,
                                                               int64_t* __restrict__ perm, int64_t* __restrict__ run_lo, int64_t* __restrict__ run_hi,
                                                               uint8_t* __restrict__ first) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int v = sorted[p];
  if (perm) perm[p] = (int64_t)perm32[p];
  if (first) first[p] = (p == 0 || sorted[p - 1] != v) ? 1 : 0;
  if (run_lo) {  // first position holding v
    long long lo = 0, hi = p;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (sorted[mid] < v) lo = mid + 1;
      else hi = mid;
    }
    run_lo[p] = lo;
  }
  if (run_hi) {  // one past the last position holding v
    long long lo = p + 1, hi = n;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (sorted[mid] <= v) lo = mid + 1;
      else hi = mid;
    }
    run_hi[p] = lo;
  }
}
__global__ __launch_bounds__(256) void iota_u32_kernel(unsigned* out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (unsigned)i;
}

static size_t group_ids_sort_temp_bytes(int64_t n) {
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, (const int*)nullptr, (int*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, 0u, 32u);
  return tb;
}

extern "C" size_t tgmx_group_ids_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return up((size_t)n * 4) * 3 + up(group_ids_sort_temp_bytes(n)) + 256;  // iota | perm32 | sorted (when the caller wants none) | sort temp
}

extern "C" int tgmx_group_ids_large(const int32_t* ids, int64_t n, int32_t* sorted, int64_t* perm, int64_t* run_lo, int64_t* run_hi,
                                    uint8_t* first, void* workspace, size_t workspace_bytes, tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0 && n < (1ll << 31), "group_ids_large: n=%lld", (long long)n);
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(ids && workspace && ((uintptr_t)workspace & 255) == 0 && workspace_bytes >= tgmx_group_ids_workspace_bytes(n),
               "group_ids_large: null / misaligned / short workspace");
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  char* base = reinterpret_cast<char*>(workspace);
  unsigned* iota = reinterpret_cast<unsigned*>(base);
  unsigned* perm32 = reinterpret_cast<unsigned*>(base + up((size_t)n * 4));
  int32_t* sorted_ws = reinterpret_cast<int32_t*>(base + 2 * up((size_t)n * 4));
  void* temp = base + 3 * up((size_t)n * 4);
  int32_t* out_sorted = sorted ? sorted : sorted_ws;
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(iota_u32_kernel, dim3(blocks), dim3(256), 0, st, iota, (long long)n);
  size_t tb = group_ids_sort_temp_bytes(n);
  const hipError_t err = rocprim::radix_sort_pairs(temp, tb, (const int*)ids, (int*)out_sorted, (const unsigned*)iota, perm32, (size_t)n, 0u, 32u, st);
  if (err != hipSuccess) {
    set_error("group_ids_large: radix sort failed: %s", hipGetErrorString(err));
    return TGMX_E_LAUNCH;
  }
  hipLaunchKernelGGL(group_ids_finish_kernel, dim3(blocks), dim3(256), 0, st, out_sorted, perm32, (long long)n, perm, run_lo, run_hi, first);
  TGMX_CHECK_LAUNCH("group_ids_large");
  return TGMX_OK;
}

extern "C" int tgmx_tgn_commit_assoc(const int32_t* src, const int32_t* dst, int64_t n, const int64_t* assoc, int64_t stamp, const float* val,
                                     const int64_t* lu, int32_t M, int32_t num_nodes, float* memory, int64_t* last_update, int32_t* status,
                                     tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0 && M > 0 && num_nodes > 0, "tgn_commit_assoc: bad sizes");
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(src && dst && assoc && val && lu && memory && last_update && status, "tgn_commit_assoc: null pointer");
  hipLaunchKernelGGL(tgn_commit_assoc_kernel, dim3((unsigned)((2 * n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n, assoc,
                     (long long)stamp, val, lu, M, num_nodes, memory, last_update, status);
  TGMX_CHECK_LAUNCH("tgn_commit_assoc");
  return TGMX_OK;
}

extern "C" int tgmx_tgn_store_batch(const int32_t* src, const int32_t* dst, const int64_t* t, const float* raw, int32_t D, int32_t n,
                                    int64_t base, int32_t* log_other, int64_t* log_t, float* log_raw, int64_t* st_lo_s, int32_t* st_cnt_s,
                                    int64_t* st_lo_d, int32_t* st_cnt_d, tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0 && n <= 1024 && D >= 0 && base >= 0, "tgn_store_batch: n=%d (at most 1024 events per call), D=%d", n, D);
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(src && dst && t && (D == 0 || (raw && log_raw)) && log_other && log_t && st_lo_s && st_cnt_s && st_lo_d && st_cnt_d,
               "tgn_store_batch: null pointer");
  StoreBatchArgs a{src, dst, t, raw, log_other, log_t, log_raw, {st_lo_s, st_lo_d}, {st_cnt_s, st_cnt_d}, base, n, D};
  int P = 64;
  while (P < n) P <<= 1;
  hipLaunchKernelGGL(tgn_store_batch_kernel, dim3(2), dim3(P), 0, (hipStream_t)stream, a);
  TGMX_CHECK_LAUNCH("tgn_store_batch");
  return TGMX_OK;
}

static int tgn_edge_list_impl(const int32_t* seed, const int32_t* nbr, const int64_t* nbr_t, const float* nbr_x, const int32_t* nbr_eid,
                              const float* table, int64_t S, int32_t k, int32_t D, const int32_t* uniq, int64_t U, const int64_t* uniq_count,
                              int64_t cap, int64_t* row_off, int64_t* edge_index, int64_t* edge_t, float* edge_x, int64_t* count,
                              tgmx_stream_t stream, bool scan_done = false) {
  TGMX_REQUIRE(S >= 0 && k > 0 && k <= 64 && D >= 0 && U >= 0 && cap >= S * k, "tgn_edge_list: bad sizes S=%lld k=%d cap=%lld", (long long)S, k,
               (long long)cap);
  TGMX_REQUIRE(count && row_off, "tgn_edge_list: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (S == 0) {
    (void)hipMemsetAsync(count, 0, sizeof(int64_t), st);
    return TGMX_OK;
  }
  TGMX_REQUIRE(seed && nbr && nbr_t && (D == 0 || ((nbr_x || (nbr_eid && table)) && edge_x)) && uniq && edge_index && edge_t, "tgn_edge_list: null pointer");
  if (!scan_done) hipLaunchKernelGGL(edge_list_scan_kernel, dim3(1), dim3(1024), 0, st, nbr, (long long)S, k, row_off, count);
  EdgeListArgs a{seed, nbr, nbr_t, nbr_x, nbr_x ? nullptr : nbr_eid, table, uniq, uniq_count, row_off, edge_index, edge_t, edge_x, S, U, cap, k, D};
  hipLaunchKernelGGL(edge_list_write_kernel, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, st, a);
  TGMX_CHECK_LAUNCH("tgn_edge_list");
  return TGMX_OK;
}

extern "C" int tgmx_tgn_edge_list(const int32_t* seed, const int32_t* nbr, const int64_t* nbr_t, const float* nbr_x, int64_t S, int32_t k,
                                  int32_t D, const int32_t* uniq, int64_t U, const int64_t* uniq_count, int64_t cap, int64_t* row_off,
                                  int64_t* edge_index, int64_t* edge_t, float* edge_x, int64_t* count, tgmx_stream_t stream) {
  return tgn_edge_list_impl(seed, nbr, nbr_t, nbr_x, nullptr, nullptr, S, k, D, uniq, U, uniq_count, cap, row_off, edge_index, edge_t, edge_x, count, stream);
}

extern "C" int tgmx_tgn_edge_list_by_id(const int32_t* seed, const int32_t* nbr, const int64_t* nbr_t, const int32_t* nbr_eid, const float* edge_table,
                                        int64_t S, int32_t k, int32_t D, const int32_t* uniq, int64_t U, const int64_t* uniq_count, int64_t cap,
                                        int64_t* row_off, int64_t* edge_index, int64_t* edge_t, float* edge_x, int64_t* count, tgmx_stream_t stream) {
  return tgn_edge_list_impl(seed, nbr, nbr_t, nullptr, nbr_eid, edge_table, S, k, D, uniq, U, uniq_count, cap, row_off, edge_index, edge_t, edge_x, count,
                            stream);
}

int tgmx_internal_edge_list(const int32_t* seed, const int32_t* nbr, const int64_t* nbr_t, const float* nbr_x, const int32_t* nbr_eid, const float* table,
                            int64_t S, int32_t k, int32_t D, const int32_t* uniq, int64_t U, const int64_t* uniq_count, int64_t cap, int64_t* row_off,
                            int64_t* edge_index, int64_t* edge_t, float* edge_x, int64_t* count, bool scan_done, tgmx_stream_t stream) {
  return tgn_edge_list_impl(seed, nbr, nbr_t, nbr_x, nbr_eid, table, S, k, D, uniq, U, uniq_count, cap, row_off, edge_index, edge_t, edge_x, count, stream,
                            scan_done && S > 0);
}

extern "C" size_t tgmx_tgn_compact_workspace_bytes(int32_t num_nodes) {
  if (num_nodes <= 0) return 0;
  size_t tb = 0;
  CompactCount cc{nullptr, nullptr, num_nodes};
  (void)rocprim::exclusive_scan(nullptr, tb, rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), cc), (long long*)nullptr, 0ll,
                                (size_t)2 * num_nodes, rocprim::plus<long long>());
  return ((size_t)2 * num_nodes * sizeof(long long) + 255) / 256 * 256 + tb + 256;
}

extern "C" int tgmx_tgn_compact(int64_t* st_lo_s, const int32_t* st_cnt_s, int64_t* st_lo_d, const int32_t* st_cnt_d, int32_t num_nodes,
                                const int32_t* old_other, const int64_t* old_t, const float* old_raw, int32_t D, int32_t* new_other, int64_t* new_t,
                                float* new_raw, void* workspace, size_t workspace_bytes, tgmx_stream_t stream) {
  TGMX_REQUIRE(num_nodes > 0 && D >= 0, "tgn_compact: bad sizes");
  TGMX_REQUIRE(st_lo_s && st_cnt_s && st_lo_d && st_cnt_d && old_other && old_t && (D == 0 || (old_raw && new_raw)) && new_other && new_t && workspace,
               "tgn_compact: null pointer");
  TGMX_REQUIRE(workspace_bytes >= tgmx_tgn_compact_workspace_bytes(num_nodes), "tgn_compact: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* base = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  long long* pos = reinterpret_cast<long long*>(base);
  char* temp = base + ((size_t)2 * num_nodes * sizeof(long long) + 255) / 256 * 256;
  size_t tb = workspace_bytes - (size_t)(temp - reinterpret_cast<char*>(workspace));
  CompactCount cc{st_cnt_s, st_cnt_d, num_nodes};
  if (rocprim::exclusive_scan(temp, tb, rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), cc), pos, 0ll, (size_t)2 * num_nodes,
                              rocprim::plus<long long>(), st) != hipSuccess) {
    set_error("tgn_compact: scan failed");
    return TGMX_E_LAUNCH;
  }
  hipLaunchKernelGGL(tgn_compact_move_kernel, dim3((unsigned)((2ll * num_nodes + 255) / 256)), dim3(256), 0, st, st_lo_s, st_cnt_s, st_lo_d, st_cnt_d,
                     (int)num_nodes, pos, old_other, old_t, old_raw, D, new_other, new_t, new_raw);
  TGMX_CHECK_LAUNCH("tgn_compact");
  return TGMX_OK;
}

// ---- one call per module forward (inference / no-grad paths): the launches of TGNMemory's look-ahead and of
// GraphAttentionEmbedding as C++ sequences -- the same entry points, in the same order, as tgm_amd/nn/tgn.py composes
// them one ctypes call at a time (a Python-mediated launch costs the host ~8 us, a C++ one ~4; cfg 3 is host-bound).
static int memory_forward_aggregate(const tgmx_tgn_memory_fwd_t* a, tgmx_stream_t stream) {
  const int64_t R = a->R;
  const int M = a->M;
  TGMX_REQUIRE(a->ws_aggr && a->ws_h && a->ws_gi && a->ws_gh && a->out_mem && a->out_lu && a->W_ih && a->W_hh, "tgn_memory_forward: null pointer");
  // (the aggregation's launch also writes the hidden-state rows h = memory[nodes]: one launch less than a separate gather)
  return tgn_aggregate_impl(a->nodes, R, a->memory, a->last_update, M, a->num_nodes, a->st_lo_s, a->st_cnt_s, a->st_lo_d, a->st_cnt_d,
                            a->log_other, a->log_t, a->log_raw, a->D, a->tw, a->tb, a->T, a->mean, a->ws_aggr, a->out_lu, a->assoc, a->stamp,
                            a->ws_h, stream);
}

// GRUCell: gi = aggr W_ih^T + b_ih, gh = h W_hh^T + b_hh (one launch for the two: they share nothing but the stream), then the gates
static int memory_forward_gru_gemms(const tgmx_tgn_memory_fwd_t* a, tgmx_stream_t stream) {
  const int64_t R = a->R;
  const int M = a->M, W = 2 * a->M + a->D + a->T;
  const GemmCall gi{a->ws_aggr, W, a->W_ih, W, a->ws_gi, 3 * M, R, 3 * M, W, a->b_ih, 0, 1, 0, 0, 0};
  const GemmCall gh{a->ws_h, M, a->W_hh, M, a->ws_gh, 3 * M, R, 3 * M, M, a->b_hh, 0, 1, 0, 0, 0};
  return tgmx_internal_sgemm_nt_pair(gi, gh, stream);
}

static int memory_forward_gru_gates(const tgmx_tgn_memory_fwd_t* a, tgmx_stream_t stream) {
  return tgmx_tgn_gru_gate(a->ws_gi, a->ws_gh, a->ws_h, a->M, a->R, a->out_mem, stream);
}

static int memory_forward_gru(const tgmx_tgn_memory_fwd_t* a, tgmx_stream_t stream) {
  if (int rc = memory_forward_gru_gemms(a, stream)) return rc;
  return memory_forward_gru_gates(a, stream);
}

extern "C" int tgmx_tgn_memory_forward(const tgmx_tgn_memory_fwd_t* a, tgmx_stream_t stream) {
  TGMX_REQUIRE(a, "tgn_memory_forward: null argument block");
  if (a->R == 0) return TGMX_OK;
  if (int rc = memory_forward_aggregate(a, stream)) return rc;
  return memory_forward_gru(a, stream);
}

// edge encoding + histogram, then the grouping of the edge ids by target (counting grouping): two launches for U <= kGroupFusedMaxU targets
// (tconv_group_scan_place_kernel), three beyond.  Returns whether the fused kernel ran: the attention then clears the histogram (count_zero).
// TGMX_TCONV_GROUP_FUSED=0: always three (A/B).
static bool launch_edge_grouping(const tgmx_tconv_fwd_t* a, hipStream_t st, const StoreBatchArgs* store = nullptr, bool* stored = nullptr) {
  const int64_t U = a->U, E = a->E;
  const int Wd = a->T + a->D;
  const char* e = getenv("TGMX_TCONV_GROUP_FUSED");  // (read per call, like TGMX_TCONV_COUNTING: the tests switch it)
  const bool fused = !(e && e[0] == '0') && U <= kGroupFusedMaxU;
  long long blocks = (E * Wd + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tconv_edge_attr_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a->last_update_local, a->src, a->t, a->msg, a->tw, a->tb,
                     a->T, a->D, (long long)E, a->edge_attr, a->tgt, a->tgt_count, (long long)U, a->status, fused ? a->cursor : (int64_t*)nullptr);
  if (fused) {
    StoreBatchArgs none{};
    const bool ride = store && store->n > 0;
    hipLaunchKernelGGL(tconv_group_scan_place_kernel, dim3((unsigned)((E + kGroupFusedThreads - 1) / kGroupFusedThreads) + (ride ? 2u : 0u)),
                       dim3(kGroupFusedThreads), 0, st, a->tgt, (long long)E, (int)U, a->tgt_count, a->cursor, a->seg_lo, a->seg_hi, a->order,
                       ride ? *store : none);
    if (ride && stored) *stored = true;
  } else {
    hipLaunchKernelGGL(tconv_group_scan_kernel, dim3(1), dim3(1024), 0, st, a->tgt_count, (long long)U, a->seg_lo, a->seg_hi, a->cursor);
    hipLaunchKernelGGL(tconv_group_place_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, a->tgt, (long long)E, (long long)U, a->cursor,
                       a->order);
  }
  return fused;
}

extern "C" int tgmx_tconv_forward(const tgmx_tconv_fwd_t* a, tgmx_stream_t stream) {
  TGMX_REQUIRE(a, "tconv_forward: null argument block");
  const int64_t U = a->U, E = a->E;
  if (U == 0) return TGMX_OK;
  const int HC = a->H * a->C, Wd = a->T + a->D;
  TGMX_REQUIRE(a->x && a->W4 && a->b4 && a->qkvs, "tconv_forward: null pointer");
  // query / key / value / skip projections of the same x: ONE batched problem over the stacked weights
  const GemmCall proj{a->x, a->in_ch, a->W4, a->in_ch, a->qkvs, HC, U, HC, a->in_ch, a->b4, 0, 4, 0, (long long)HC * a->in_ch, U * HC};
  auto run = [&](const GemmCall& c) {
    return tgmx_sgemm_nt(c.A, c.lda, c.B, c.ldb, c.C, c.ldc, c.M, c.N, c.K, c.bias, c.relu, c.batch, c.sA, c.sB, c.sC, stream);
  };
  int rc = 0;
  if (E == 0) return run(proj);
  TGMX_REQUIRE(a->src && a->tgt && a->t && a->edge_attr && a->eproj && a->order && a->seg_lo && a->seg_hi && a->status, "tconv_forward: null pointer");
  float* out = a->qkvs + 3 * U * HC;  // the skip projection; the attention output is added to it
  const char* knob = getenv("TGMX_TCONV_COUNTING");  // A/B knob, read per call (the tests switch it)
  const bool count_off = knob && atoi(knob) == 0;
  if (a->tgt_count && a->cursor && a->order_big && !count_off) {
    // Counting grouping: the edge encoding's launch also counts the edges per target, one workgroup scans the counts into the segments,
    // one small launch places every edge id with an atomic cursor, and the attention sorts a segment (1-3 ids for most targets) before
    // walking it -- ascending edge id is the order a stable sort by target gives, so the result equals the sorted path's bit for bit.
    // 3 launches less work than: keys + 4-5 radix-sort launches + the library's copy-back + bounds (~50 us of a 264 us batch, round 4).
    hipStream_t st = (hipStream_t)stream;
    const bool fused_group = launch_edge_grouping(a, st);
    TGMX_CHECK_LAUNCH("tconv_forward(grouping)");
    // the node projections (x -> q, k, v, skip) and the edge projection (edge_attr -> eproj) do not depend on each other: one launch
    const GemmCall eproj{a->edge_attr, Wd, a->W_edge, Wd, a->eproj, HC, E, HC, Wd, nullptr, 0, 1, 0, 0, 0};
    if ((rc = tgmx_internal_sgemm_nt_pair(proj, eproj, stream))) return rc;
    TconvArgs t{a->qkvs, a->qkvs + U * HC, a->qkvs + 2 * U * HC, a->eproj, a->order, a->src, a->seg_lo, a->seg_hi, out, U, a->H, a->C,
                1.0f / sqrtf((float)a->C)};
    t.drop = make_dropout(nullptr);
    t.unsorted = 1;
    t.order_big = a->order_big;
    t.short_ok = tconv_short_ok(t);
    t.count_zero = fused_group ? a->tgt_count : nullptr;
    hipLaunchKernelGGL(tconv_attend_kernel, dim3((unsigned)U), dim3(256), 0, st, t);
    TGMX_CHECK_LAUNCH("tconv_attend");
    return TGMX_OK;
  }
  TGMX_REQUIRE(a->sort_ws, "tconv_forward: null pointer");
  if ((rc = run(proj))) return rc;
  if ((rc = tgmx_tconv_edge_attr(a->last_update_local, a->src, a->t, a->msg, a->tw, a->tb, a->T, a->D, E, a->edge_attr, stream))) return rc;
  if ((rc = tgmx_sgemm_nt(a->edge_attr, Wd, a->W_edge, Wd, a->eproj, HC, E, HC, Wd, nullptr, 0, 1, 0, 0, 0, stream))) return rc;
  if ((rc = tgmx_segment_sort(a->tgt, E, (int32_t)U, a->order, a->seg_lo, a->seg_hi, a->sort_ws, a->sort_ws_bytes, a->status, stream))) return rc;
  return tgmx_tconv_attend(a->qkvs, a->qkvs + U * HC, a->qkvs + 2 * U * HC, a->eproj, a->order, a->src, a->seg_lo, a->seg_hi, U, a->H, a->C,
                           1.0f / sqrtf((float)a->C), out, nullptr, stream);
}

// the model side of one TGN batch: memory forward -> embedding -> update_state (commit of the rows just computed + the batch's store)
namespace {
struct StepSide {  // a stream + two events of the library's own, per calling thread: the edge-side chain of a step runs beside the node-side chain
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, join_store = nullptr;
  int device = -1;
  bool failed = false;
};
StepSide* step_side() {
  static thread_local StepSide s;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (s.stream && s.device == dev) return &s;
  if (s.stream) return nullptr;  // (one device per thread: anything else keeps the single-stream order)
  if (s.failed) return nullptr;  // (a failed set-up is not retried on every call: the step then keeps the single-stream order)
  if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) {
    s.stream = nullptr;
    s.failed = true;
    return nullptr;
  }
  if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&s.join_store, hipEventDisableTiming) != hipSuccess) {
    // partial set-up: give back what was created
    if (s.fork) (void)hipEventDestroy(s.fork);
    if (s.join) (void)hipEventDestroy(s.join);
    if (s.join_store) (void)hipEventDestroy(s.join_store);
    (void)hipStreamDestroy(s.stream);
    s.stream = nullptr;
    s.fork = s.join = s.join_store = nullptr;
    s.failed = true;
    return nullptr;
  }
  s.device = dev;
  return &s;
}
// After the fork the library's stream may hold kernels that read the step's buffers: whatever happens next, the caller's stream is made to wait
// for them before tgmx_tgn_step returns (the caller frees / reuses the buffers in stream order of ITS stream only).
struct JoinOnExit {
  hipStream_t caller, side;
  hipEvent_t join;
  bool armed = false, joined = false;
  ~JoinOnExit() {
    if (armed && !joined) {
      if (hipEventRecord(join, side) != hipSuccess || hipStreamWaitEvent(caller, join, 0) != hipSuccess) (void)hipStreamSynchronize(side);
    }
  }
};
}  // namespace

extern "C" int tgmx_tgn_step(const tgmx_tgn_step_t* a, tgmx_stream_t stream) {
  TGMX_REQUIRE(a && a->mem, "tgn_step: null argument block");
  const tgmx_tgn_memory_fwd_t* m = a->mem;
  TGMX_REQUIRE(a->n >= 0 && a->n <= 1024, "tgn_step: n=%d (at most 1024 events per call)", a->n);
  TGMX_REQUIRE(m->assoc && a->memory && a->last_update && a->reuse_status, "tgn_step: the commit needs mem->assoc / memory / last_update / reuse_status");
  int rc = TGMX_OK;
  bool stored = false;  // the batch's message store has been issued (on the library's stream, beside the node side)
  bool committed = false;  // the commit has been issued (as extra workgroups of the attention's launch)
  const tgmx_tconv_fwd_t* c = a->conv;
  // Two chains hang off the aggregation: the NODE side (GRU GEMMs, gates, the q / k / v / skip projections of the new memory rows) and the
  // EDGE side (edge encoding from the new last_update rows, the grouping by target, the edge projection).  They meet at the attention.
  // The edge side -- four small launches, ~27 us -- runs on a stream of the library's own beside the node side's two GEMM launches; one
  // fork and one join event per step.  (Round 4 measured this with the host as the bottleneck and lost; with the step issued from C and
  // the loader on its own stream the host has ~65 us of slack per batch.)  TGMX_TGN_STEP_OVERLAP=0: everything on the caller's stream.
  static const bool overlap_knob = [] { const char* e = getenv("TGMX_TGN_STEP_OVERLAP"); return !(e && e[0] == '0'); }();
  const char* knob = getenv("TGMX_TCONV_COUNTING");
  const bool counting = c && c->E > 0 && c->U > 0 && c->tgt_count && c->cursor && c->order_big && !(knob && atoi(knob) == 0);
  StepSide* side = (overlap_knob && counting && m->R > 0) ? step_side() : nullptr;
  if (!side) {
    if ((rc = tgmx_tgn_memory_forward(m, stream))) return rc;
    if (c && (rc = tgmx_tconv_forward(c, stream))) return rc;
  } else {
    hipStream_t st = (hipStream_t)stream, ss = side->stream;
    const int64_t U = c->U, E = c->E;
    const int HC = c->H * c->C, Wd = c->T + c->D;
    TGMX_REQUIRE(c->x && c->W4 && c->b4 && c->qkvs && c->src && c->tgt && c->t && c->edge_attr && c->eproj && c->order && c->seg_lo && c->seg_hi && c->status,
                 "tgn_step: null pointer in the embedding's argument block");
    if ((rc = memory_forward_aggregate(m, stream))) return rc;
    if (hipEventRecord(side->fork, st) != hipSuccess) {
      set_error("tgn_step: fork failed");
      return TGMX_E_LAUNCH;
    }
    // The node side's first launch -- the GRU's two GEMMs, the longest kernel of the step (30-40 us) -- goes out BEFORE the edge side's five:
    // issued behind them, the caller's stream sat idle for the ~25 us the host needs to issue those (kernel trace, round 6); its remaining
    // two launches follow the edge side and are queued long before the GEMMs finish.  TGMX_TGN_NODE_FIRST=0: the edge side first (A/B).
    const char* nf = getenv("TGMX_TGN_NODE_FIRST");
    const bool node_first = !(nf && nf[0] == '0');
    if (node_first && (rc = memory_forward_gru_gemms(m, stream))) return rc;  // (nothing on the library's stream yet: a plain return)
    if (hipStreamWaitEvent(ss, side->fork, 0) != hipSuccess) {  // (the other stream's wait is issued behind the GEMMs' launch: they start sooner)
      set_error("tgn_step: fork failed");
      return TGMX_E_LAUNCH;
    }
    JoinOnExit guard{st, ss, side->join};
    guard.armed = true;  // from here on every return joins the library's stream back into the caller's
    // ---- edge side, on the library's stream ----
    // the batch's message store rides the grouping's launch when that is the fused one (TGMX_TGN_STORE_RIDER=0: its own launch, A/B)
    static const bool store_side = [] { const char* e = getenv("TGMX_TGN_STORE_SIDE"); return !(e && e[0] == '0'); }();
    const char* sr = getenv("TGMX_TGN_STORE_RIDER");
    StoreBatchArgs sargs{a->src, a->dst, a->t, a->raw, a->log_other, a->log_t, a->log_raw, {a->st_lo_s, a->st_lo_d}, {a->st_cnt_s, a->st_cnt_d}, a->log_base,
                         a->n, m->D};
    const bool store_rides = store_side && a->n > 0 && !(sr && sr[0] == '0') && a->src && a->dst && a->t && (m->D == 0 || (a->raw && a->log_raw)) &&
                             a->log_other && a->log_t && a->st_lo_s && a->st_cnt_s && a->st_lo_d && a->st_cnt_d && a->log_base >= 0;
    const bool fused_group = launch_edge_grouping(c, ss, store_rides ? &sargs : nullptr, &stored);
    TGMX_CHECK_LAUNCH("tgn_step(grouping)");
    if ((rc = tgmx_sgemm_nt(c->edge_attr, Wd, c->W_edge, Wd, c->eproj, HC, E, HC, Wd, nullptr, 0, 1, 0, 0, 0, (tgmx_stream_t)ss))) return rc;
    // The batch's message store (tgn.py:173,176: one latency-bound workgroup per role, ~12 us) only needs the aggregation above to be done
    // with the OLD windows -- it is behind the fork -- and the next step's aggregation to come after it -- the join covers that: it runs here,
    // beside the node side's GEMMs, instead of at the end of the caller's stream.  TGMX_TGN_STORE_SIDE=0: behind the commit, as before (A/B).
    // TGMX_TGN_STORE_JOIN_LATE=1 (A/B, off by default): two joins -- the attention waits for the edge projection only, the caller's stream
    // picks up the store behind the attention's launch (the store is the side chain's last and longest launch, and the attention needs
    // nothing of it).  Measured over 16 alternating segments: 137.4 us per batch with two joins, 134.9 with one -- the second record + wait cost
    // the host more than the attention gains on a step whose host side is the longer one.
    const char* jl = getenv("TGMX_TGN_STORE_JOIN_LATE");
    const bool two_joins = store_side && a->n > 0 && jl && jl[0] == '1';
    if (two_joins && hipEventRecord(side->join, ss) != hipSuccess) {
      set_error("tgn_step: join record failed");
      return TGMX_E_LAUNCH;
    }
    if (store_side && a->n > 0 && !stored) {
      if ((rc = tgmx_tgn_store_batch(a->src, a->dst, a->t, a->raw, m->D, a->n, a->log_base, a->log_other, a->log_t, a->log_raw, a->st_lo_s, a->st_cnt_s,
                                     a->st_lo_d, a->st_cnt_d, (tgmx_stream_t)ss)))
        return rc;
      stored = true;
    }
    if (hipEventRecord(two_joins ? side->join_store : side->join, ss) != hipSuccess) {
      set_error("tgn_step: join record failed");
      return TGMX_E_LAUNCH;
    }
    // ---- node side, on the caller's stream ----
    if (!node_first && (rc = memory_forward_gru_gemms(m, stream))) return rc;
    if ((rc = memory_forward_gru_gates(m, stream))) return rc;
    if ((rc = tgmx_sgemm_nt(c->x, c->in_ch, c->W4, c->in_ch, c->qkvs, HC, U, HC, c->in_ch, c->b4, 0, 4, 0, (int64_t)HC * c->in_ch, U * HC, stream))) return rc;
    if (hipStreamWaitEvent(st, side->join, 0) != hipSuccess) {
      set_error("tgn_step: join failed");
      return TGMX_E_LAUNCH;
    }
    guard.joined = !two_joins;  // (two joins: the store is still out; an error return from here on takes the guard's record + wait)
    float* out = c->qkvs + 3 * U * HC;
    TconvArgs t{c->qkvs, c->qkvs + U * HC, c->qkvs + 2 * U * HC, c->eproj, c->order, c->src, c->seg_lo, c->seg_hi, out, U, c->H, c->C,
                1.0f / sqrtf((float)c->C)};
    t.drop = make_dropout(nullptr);
    t.unsorted = 1;
    t.order_big = c->order_big;
    t.short_ok = tconv_short_ok(t);
    t.count_zero = fused_group ? c->tgt_count : nullptr;
    // the commit of the look-ahead rows rides the attention's launch as ceil(2n / 4) extra workgroups (CommitRider: it depends on the gates'
    // launch, two back on this stream, and on nothing of the attention's).  TGMX_TGN_COMMIT_RIDER=0: its own launch behind the attention (A/B)
    const char* rider_knob = getenv("TGMX_TGN_COMMIT_RIDER");  // (read per call: the tests switch it)
    unsigned extra = 0;
    if (!(rider_knob && rider_knob[0] == '0') && a->n > 0) {
      t.commit = CommitRider{a->src, a->dst, m->assoc, m->out_mem, m->out_lu, a->memory, a->last_update, a->reuse_status, a->n, (long long)m->stamp, m->M,
                             m->num_nodes};
      extra = (unsigned)((2 * a->n + 3) / 4);
      committed = true;
    }
    hipLaunchKernelGGL(tconv_attend_kernel, dim3((unsigned)U + extra), dim3(256), 0, st, t);
    TGMX_CHECK_LAUNCH("tgn_step(attend)");
    if (two_joins) {
      if (hipStreamWaitEvent(st, side->join_store, 0) != hipSuccess) {
        set_error("tgn_step: join failed");
        return TGMX_E_LAUNCH;
      }
      guard.joined = true;
    }
  }
  if (a->n == 0) return TGMX_OK;
  if (committed) {
    if (stored) return TGMX_OK;
    return tgmx_tgn_store_batch(a->src, a->dst, a->t, a->raw, m->D, a->n, a->log_base, a->log_other, a->log_t, a->log_raw, a->st_lo_s, a->st_cnt_s,
                                a->st_lo_d, a->st_cnt_d, stream);
  }
  if (stored) {
    return tgmx_tgn_commit_assoc(a->src, a->dst, a->n, m->assoc, m->stamp, m->out_mem, m->out_lu, m->M, m->num_nodes, a->memory, a->last_update,
                                 a->reuse_status, stream);
  }
  if ((rc = tgmx_tgn_commit_assoc(a->src, a->dst, a->n, m->assoc, m->stamp, m->out_mem, m->out_lu, m->M, m->num_nodes, a->memory, a->last_update,
                                  a->reuse_status, stream)))
    return rc;
  return tgmx_tgn_store_batch(a->src, a->dst, a->t, a->raw, m->D, a->n, a->log_base, a->log_other, a->log_t, a->log_raw, a->st_lo_s, a->st_cnt_s,
                              a->st_lo_d, a->st_cnt_d, stream);
}
