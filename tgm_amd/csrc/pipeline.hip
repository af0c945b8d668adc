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

extern "C" int tgmx_pipeline_step(const tgmx_pipeline_t* p, int64_t edge_lo, int64_t n_edges, uint64_t neg_call,
                                  const tgmx_pipeline_out_t* out, tgmx_stream_t stream) {
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
  return tgmx_recency_step(&s, stream);
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
