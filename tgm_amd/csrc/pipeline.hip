// The loader's per-batch call as one entry point (include/tgm_amd.h: tgmx_pipeline_step) and the host-side slicer.
//
// Why this exists: at the headline shape one batch is ~40 us of kernels, and composing it from Python (slice view,
// materialize, hook manager, one hook call per stage, a torch allocation per output) costs more than that on the
// host.  Here the whole chain is argument arithmetic in C++ around tgmx_recency_step: no kernel of its own -- the
// rank's share is pointer offsets, the negatives are drawn inside the seed fetch of the lookup kernels, the outputs
// are the caller's preallocated pool slot.
#include <algorithm>

#include "common.h"

using namespace tgmx;

// dev[0 .. 3) -> the pinned host mirror (unique count | status | edge count): system-scope stores from three lanes
__global__ __launch_bounds__(64) void sizes_to_host_kernel(const int64_t* __restrict__ dev, int64_t* __restrict__ host) {
  if (threadIdx.x < 3) __hip_atomic_store(&host[threadIdx.x], dev[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int pipeline_post(const tgmx_pipeline_t* p, const tgmx_recency_step_t& s, int64_t edge_lo, int64_t n_edges, long long share,
                         const tgmx_pipeline_out_t* out, const tgmx_pipeline_post_t* post, tgmx_stream_t stream) {
  TGMX_REQUIRE(post->dev_sizes && post->host_sizes && post->sizes_ready, "pipeline_step: post block needs dev_sizes / host_sizes / sizes_ready");
  long long rows[TGMX_MAX_HOPS + 1];  // rows[h] = seeds of hop h
  rows[0] = share * p->n_roles;
  for (int h = 0; h < s.n_hops; ++h) rows[h + 1] = rows[h] * s.k[h];
  bool scan_rides = false;
  if (post->dedup) {
    TGMX_REQUIRE(post->dedup_ws && post->uniq_out, "pipeline_step: null dedup buffer");
    const int32_t* parts[16];
    int64_t sizes[16];
    int np = 0;
    parts[np] = p->src + edge_lo; sizes[np++] = n_edges;
    parts[np] = p->dst + edge_lo; sizes[np++] = n_edges;
    if (post->dedup_neg && s.neg_out) { parts[np] = s.neg_out; sizes[np++] = share; }
    if (post->dedup_nbr)
      for (int h = 0; h < s.n_hops && np < 16; ++h) { parts[np] = s.out_nid[h]; sizes[np++] = rows[h + 1]; }
    // the edge list's row scan reads the sampler's outputs only: it rides the marking launch as one more workgroup
    // (TGMX_EDGE_SCAN_RIDE=0: its own launch, the A/B knob)
    static const bool ride_knob = [] { const char* e = getenv("TGMX_EDGE_SCAN_RIDE"); return !(e && e[0] == '0'); }();
    EdgeListScan rider{};
    if (ride_knob && post->edge_hop >= 0 && post->edge_hop < s.n_hops && rows[post->edge_hop] > 0 && post->row_off) {
      const int h = post->edge_hop;
      rider = EdgeListScan{s.out_nid[h], rows[h], s.k[h], post->row_off, post->dev_sizes + 2};
      scan_rides = true;
    }
    const int rc = tgmx_internal_unique_ids(parts, sizes, np, post->num_nodes, post->dedup_ws, post->uniq_out, post->dev_sizes,
                                            reinterpret_cast<int32_t*>(post->dev_sizes + 1), scan_rides ? &rider : nullptr, stream);
    if (rc) return rc;
  }
  if (post->edge_hop >= 0) {
    const int h = post->edge_hop;
    TGMX_REQUIRE(post->dedup && h < s.n_hops, "pipeline_step: the edge list needs the unique ids and a sampled hop");
    const int32_t* seeds = h == 0 ? out->seed_nid0 : s.out_nid[h - 1];
    // edge features by id (the lookups published edge ids, no dense copy): the rows come from the resident store where the list is written
    const bool by_id = !s.out_x[h] && s.out_eid[h] && s.D > 0;
    const int rc = tgmx_internal_edge_list(seeds, s.out_nid[h], s.out_ts[h], by_id ? nullptr : s.out_x[h], by_id ? s.out_eid[h] : nullptr,
                                           by_id ? p->edge_x : nullptr, rows[h], s.k[h], s.D, post->uniq_out, 0, post->dev_sizes, post->edge_cap,
                                           post->row_off, post->edge_index, post->edge_t, post->edge_x, post->dev_sizes + 2, scan_rides, stream);
    if (rc) return rc;
  }
  // The three sizes go to the pinned mirror with ONE wave's stores (host-coherent memory, visible when the event behind the launch completes)
  // instead of a device -> host copy: the runtime's blit kernel took 12.7 us on the loader's chain for 24 bytes (rocprofv3, cfg 3; the
  // pipeline's time per batch does not change -- the chain is not its bound -- but the chain is 10 us shorter).  TGMX_SIZES_BLIT=1:
  // hipMemcpyAsync as before (A/B).
  static const bool blit = [] { const char* e = getenv("TGMX_SIZES_BLIT"); return e && e[0] == '1'; }();
  bool ok = true;
  if (blit) {
    ok = hipMemcpyAsync(post->host_sizes, post->dev_sizes, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess;
  } else {
    hipLaunchKernelGGL(sizes_to_host_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, post->dev_sizes, post->host_sizes);
    ok = hipGetLastError() == hipSuccess;
  }
  if (!ok || hipEventRecord((hipEvent_t)post->sizes_ready, (hipStream_t)stream) != hipSuccess) {
    set_error("pipeline_step: size read-back failed");
    return TGMX_E_LAUNCH;
  }
  return TGMX_OK;
}

extern "C" int tgmx_pipeline_step(const tgmx_pipeline_t* p, int64_t edge_lo, int64_t n_edges, uint64_t neg_call,
                                  const tgmx_pipeline_out_t* out, const tgmx_pipeline_post_t* post, tgmx_stream_t stream) {
  TGMX_REQUIRE(p && out, "pipeline_step: null argument block");
  TGMX_REQUIRE(p->src && p->dst && p->ts && p->num_edges >= 0, "pipeline_step: null stream pointer");
  TGMX_REQUIRE(edge_lo >= 0 && n_edges >= 0 && edge_lo + n_edges <= p->num_edges, "pipeline_step: edges [%lld, +%lld) outside the store (%lld)",
               (long long)edge_lo, (long long)n_edges, (long long)p->num_edges);
  TGMX_REQUIRE(p->world >= 1 && p->rank >= 0 && p->rank < p->world, "pipeline_step: rank %d of %d", p->rank, p->world);
  TGMX_REQUIRE(p->n_roles >= 1 && p->n_roles <= TGMX_MAX_SEED_GROUPS, "pipeline_step: %d seed roles", p->n_roles);
  tgmx_recency_step_t s = p->step;
  const long long lo = n_edges * p->rank / p->world, hi = n_edges * (p->rank + 1) / p->world;  // this rank's share
  const long long share = hi - lo;
  s.n_groups = p->n_roles;
  s.neg_out = nullptr;
  s.neg_group = -1;
  for (int g = 0; g < p->n_roles; ++g) {
    s.grp_ts[g] = p->ts + edge_lo + lo;
    s.grp_n[g] = share;
    switch (p->seed_role[g]) {
      case TGMX_SEED_SRC: s.grp_nid[g] = p->src + edge_lo + lo; break;
      case TGMX_SEED_DST: s.grp_nid[g] = p->dst + edge_lo + lo; break;
      case TGMX_SEED_NEG:
        TGMX_REQUIRE(s.neg_group < 0, "pipeline_step: more than one negatives role");
        TGMX_REQUIRE(share == 0 || (out->neg && out->neg_time), "pipeline_step: null negatives output");
        s.grp_nid[g] = nullptr;
        s.neg_group = g; s.neg_low = p->neg_low; s.neg_high = p->neg_high; s.neg_seed = p->neg_seed; s.neg_call = neg_call;
        s.neg_index0 = lo;  // the share's draws are a slice of the whole batch's (tgmx_recency_step_t.neg_index0)
        s.neg_out = out->neg; s.neg_time_out = out->neg_time;
        break;
      default: TGMX_REQUIRE(false, "pipeline_step: seed role %d", p->seed_role[g]);
    }
  }
  s.seed_nid0 = out->seed_nid0;
  s.seed_ts0 = out->seed_ts0;
  for (int h = 0; h < s.n_hops && h < TGMX_MAX_HOPS; ++h) {
    s.out_nid[h] = out->out_nid[h];
    s.out_ts[h] = out->out_ts[h];
    s.out_x[h] = out->out_x[h];
    s.out_valid[h] = out->out_valid[h];
    s.out_valid_prev[h] = out->out_valid_prev[h];
    s.out_eid[h] = out->out_eid[h];
  }
  s.timed_hop = out->timed_hop;
  s.ev_start = out->ev_start;
  s.ev_stop = out->ev_stop;
  if (s.indptr) {  // static index: stateless lookup of the edges before this batch (and not before the epoch's first)
    s.ev_hi = edge_lo;
    s.n = 0;
  } else if (p->update && n_edges > 0) {
    s.src = p->src + edge_lo;
    s.dst = p->dst + edge_lo;
    s.ts = p->ts + edge_lo;
    s.edge_x = (p->edge_x && s.D > 0) ? p->edge_x + edge_lo * (long long)s.D : nullptr;
    s.n = n_edges;
    s.eid0 = edge_lo;
  } else {
    s.n = 0;
  }
  if (share == 0) {
    // no seeds: the reference emits empties and SKIPS the update (recency.py:127-139)
    return TGMX_OK;
  }
  const int rc = tgmx_recency_step(&s, stream);
  if (rc || !post) return rc;
  return pipeline_post(p, s, edge_lo, n_edges, share, out, post, stream);
}

extern "C" int tgmx_slice(const int64_t* t, int64_t n, int32_t has_start_time, int64_t start_time, int32_t has_end_time,
                          int64_t end_time, int64_t start_idx, int64_t end_idx, int64_t* lb, int64_t* ub) {
  TGMX_REQUIRE((t || n == 0) && n >= 0 && lb && ub, "slice: bad arguments");
  long long l = has_start_time ? std::lower_bound(t, t + n, start_time) - t : 0;  // searchsorted(side='left')
  long long u = has_end_time ? std::upper_bound(t, t + n, end_time) - t : n;     // searchsorted(side='right') on the inclusive end
  const long long lo_c = start_idx > 0 ? start_idx : 0, hi_c = end_idx >= 0 ? end_idx : n;
  l = std::max(lo_c, std::min(hi_c, l));
  u = std::max(lo_c, std::min(hi_c, u));
  *lb = l;
  *ub = u;
  return TGMX_OK;
}

// ---- byte accounting of one lookup launch (bench.py's roofline; one launch instead of eight torch reductions inside the timed region)
namespace {
__global__ __launch_bounds__(256) void lookup_accounting_kernel(const int32_t* __restrict__ ids, long long slots, const int32_t* __restrict__ sp,
                                                                const int32_t* __restrict__ sc, long long rows, unsigned long long* __restrict__ counts) {
  unsigned long long v = 0, m = 0, c = 0;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += step) v += ids[i] != -1;
  if (sp && sc)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += step) {
      const int a = sp[i], b = sc[i];
      m += (unsigned long long)(a > b ? a : b);
      c += (unsigned long long)b;
    }
  for (int o = 32; o > 0; o >>= 1) {
    v += __shfl_xor(v, o);
    m += __shfl_xor(m, o);
    c += __shfl_xor(c, o);
  }
  if ((threadIdx.x & 63) == 0) {  // one partial triple per wave: nothing to zero beforehand, the host adds them up later
    unsigned long long* o = counts + ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 3;
    o[0] = v; o[1] = m; o[2] = c;
  }
}
}  // namespace

extern "C" int tgmx_lookup_accounting(const int32_t* ids, int64_t slots, const int32_t* span_prev, const int32_t* span_cur, int64_t rows,
                                      int64_t* counts, tgmx_stream_t stream) {
  TGMX_REQUIRE(ids && counts && slots >= 0 && rows >= 0, "lookup_accounting: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(lookup_accounting_kernel, dim3(TGMX_ACCOUNTING_PARTIALS / 4), dim3(256), 0, st, ids, (long long)slots, span_prev, span_cur, (long long)rows,
                     reinterpret_cast<unsigned long long*>(counts));
  TGMX_CHECK_LAUNCH("lookup_accounting");
  return TGMX_OK;
}
