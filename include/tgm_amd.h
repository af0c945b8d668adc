/*
 * tgm_amd.h -- C ABI of libtgm_amd.so: the MI355X (gfx950) kernels behind the
 * drop-in hooks of tgm_amd.
 *
 * The reference (tgm-team/tgm) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY.md F1), so each entry point below names the reference *Python*
 * routine it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into a contiguous buffer (the Python
 *     side passes torch.Tensor.data_ptr()); no torch types cross the boundary;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued asynchronously, nothing here synchronises;
 *   - nothing here allocates device memory: callers pass outputs/workspaces;
 *   - return value: 0 = ok, negative = TGMX_E_* (message: tgmx_last_error()).
 *   - `status` words are device int32 bit-sets written by kernels (TGMX_ST_*),
 *     so input validation costs no host round trip; the caller reads them when
 *     it chooses to (the hooks: every call by default, or deferred).
 */
#ifndef TGM_AMD_H
#define TGM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGMX_ABI_VERSION 7

#define TGMX_OK 0
#define TGMX_E_INVALID (-1)  /* bad argument (null pointer, size, alignment) */
#define TGMX_E_LAUNCH (-2)   /* hipLaunch / runtime error                    */
#define TGMX_E_UNSUPPORTED (-3)

/* bits of a device-side status word */
#define TGMX_ST_SEED_RANGE 1 /* a hop-0 seed id outside [0, num_nodes)        */
#define TGMX_ST_SEED_TIME 2  /* a hop-0 seed time < 0                         */
#define TGMX_ST_EDGE_RANGE 4 /* an edge endpoint outside [0, num_nodes)       */
#define TGMX_ST_SCRATCH 8    /* the head of the update scratch was not zero at first use */
#define TGMX_ST_TS_BOUND 16  /* a batch timestamp outside [0, ts_bound] (tgmx_recency_step_t) */

typedef void* tgmx_stream_t;
typedef void* tgmx_event_t; /* hipEvent_t */

/* One adjacency / ring record: 16 bytes, 16-byte aligned, so one lane moves
 * one record with a single dwordx4 access. */
typedef struct tgmx_adj {
  int32_t nbr; /* neighbor node id, -1 = empty slot                          */
  int32_t eid; /* row of the resident edge_x table holding the edge features */
  int64_t ts;  /* edge timestamp                                             */
} tgmx_adj_t;

int tgmx_version(void);
/* sizeof of the argument structs as this library was compiled (a binding checks its own mirror against it):
 * 0 tgmx_adj_t, 1 tgmx_recency_step_t, 2 tgmx_tgat_layer_t, 3 tgmx_tgat_model_t, 4 tgmx_tgat_hop_t, 5 tgmx_tgat_layout_t,
 * 6 tgmx_pipeline_t, 7 tgmx_pipeline_out_t, 8 tgmx_dropout_t, 9 tgmx_tgn_memory_fwd_t, 10 tgmx_tconv_fwd_t, 11 tgmx_pipeline_post_t,
 * 12 tgmx_tgn_step_t */
size_t tgmx_abi_sizeof(int32_t which);
const char* tgmx_last_error(void);

/* HIP timing events.  The lookup entry points accept an optional (ev_start,
 * ev_stop) pair which they record on `stream` immediately before / after their
 * kernel, so a caller can time exactly that launch on the stream it runs on
 * (bench.py's roofline leg).  tgmx_event_elapsed_ms synchronises on `stop`. */
int tgmx_event_create(tgmx_event_t* ev);
int tgmx_event_destroy(tgmx_event_t ev);
int tgmx_event_elapsed_ms(tgmx_event_t start, tgmx_event_t stop, float* ms);
int tgmx_event_synchronize(tgmx_event_t ev); /* host wait until the work recorded before `ev` is done */
/* ABI v7 -- ORDERING between two streams of one process (the loader's chain on a side stream beside the model's chain: DESIGN.md 3.3c,
 * DGDataLoader(side_stream=True)).  The reference has no counterpart: tgm/data/loader.py:158-170 runs hooks and model on one stream.
 * tgmx_event_create_sync: an event without timing (cheaper to record).  tgmx_event_record: everything enqueued on `stream` so far.
 * tgmx_stream_wait_event: work enqueued on `stream` from now on starts after the recorded work (no host wait).
 * tgmx_stream_handoff = record(ev, from) + wait(to, ev) in one call. */
int tgmx_event_create_sync(tgmx_event_t* ev);
int tgmx_event_record(tgmx_event_t ev, tgmx_stream_t stream);
int tgmx_stream_wait_event(tgmx_stream_t stream, tgmx_event_t ev);
int tgmx_stream_handoff(tgmx_stream_t from, tgmx_stream_t to, tgmx_event_t ev);

/* ------------------------------------------------------------------------
 * k-most-recent neighbor lookup over a static per-node index (CSR).
 * Replaces RecencyNeighborHook._get_recency_neighbors
 * (tgm/hooks/neighbors/recency.py:239-321) for a chronological loader whose
 * batch boundaries were known when the index was built.
 *
 *   indptr[num_nodes+1], adj[indptr[num_nodes]]: node n's entries, ordered by
 *     (batch_idx, time, role, eid); entries with eid in [ev_lo, ev_hi) are
 *     visible, of which only the last B are observable (B = max(num_nbrs)).
 *   edge_x [num_edges, D] row-major float32 (NULL iff D == 0)
 *   seeds[S] (int32; -1 = pad seed allowed iff allow_pad), qtimes[S]
 *   out_nid [S,k] int32 (pad -1), out_ts [S,k] int64 (pad 0),
 *   out_x [S,k,D] float32 (pad 0): oldest -> newest, right aligned.
 * ------------------------------------------------------------------------ */
int tgmx_recency_lookup_csr(const int64_t* indptr, const tgmx_adj_t* adj,
                            const float* edge_x, int32_t D,
                            const int32_t* seeds, const int64_t* qtimes, int64_t S,
                            int32_t k, int32_t B, int64_t ev_lo, int64_t ev_hi,
                            int32_t num_nodes, int32_t allow_pad,
                            int32_t* out_nid, int64_t* out_ts, float* out_x,
                            int32_t* status, tgmx_stream_t stream,
                            tgmx_event_t ev_start, tgmx_event_t ev_stop);

/* Uniform neighbor sampling, NeighborSamplerHook (tgm/hooks/neighbors/uniform.py:87-142) over
 * DGStorageArrayBackend.get_nbrs (tgm/core/_storage/backends/array_backend.py:108-171), against the static index built
 * with num_batches = -1.  Candidates of node n = its entries with eid < ev_hi (the hook passes the first edge whose time
 * reaches the batch's earliest timestamp: "strictly before this batch").  At most k candidates: all of them, in event
 * order, left aligned, padded with (-1, 0, 0.0) -- bit-exact with the reference.  More than k: a uniformly random
 * k-permutation without replacement (virtual Fisher-Yates), a deterministic function of (rng_seed, rng_stream, node) so
 * that every occurrence of a node in one call gets the same row, as in the reference.  The reference draws with
 * Python's Mersenne Twister (random.sample), so parity there is distributional.  k <= 64. */
int tgmx_uniform_lookup_csr(const int64_t* indptr, const tgmx_adj_t* adj, const float* edge_x, int32_t D,
                            const int32_t* seeds, int64_t S, int32_t k, int64_t ev_hi, int32_t num_nodes,
                            int32_t allow_pad, uint64_t rng_seed, uint64_t rng_stream,
                            int32_t* out_nid, int64_t* out_ts, float* out_x, int32_t* status,
                            tgmx_stream_t stream);

/* ------------------------------------------------------------------------
 * Streaming mode: per-node rings of B records, the exact state machine of the
 * reference hook (recency.py:93-102 state, :239-321 lookup, :323-399 update).
 *   ring   [num_nodes, B] records (init nbr=-1, eid=0, ts=0), write_pos[num_nodes]
 *   ring_x [num_nodes, B, D] feature row of every slot (need not be zeroed:
 *          rows of empty slots are never read)
 * ------------------------------------------------------------------------ */
int tgmx_ring_lookup(const tgmx_adj_t* ring, const int32_t* write_pos,
                     const float* ring_x, int32_t D,
                     const int32_t* seeds, const int64_t* qtimes, int64_t S,
                     int32_t k, int32_t B, int32_t num_nodes, int32_t allow_pad,
                     int32_t* out_nid, int64_t* out_ts, float* out_x,
                     int32_t* status, tgmx_stream_t stream,
                     tgmx_event_t ev_start, tgmx_event_t ev_stop);

/* Append one batch of n edges (src[i], dst[i], ts[i], edge_x[i, :]) to the rings
 * exactly as recency.py:323-399 does: stable sort of the cat[src-role, dst-role]
 * entries by key = node * (max_t + 1) + t, per-run "keep last B", scatter at
 * (write_pos + rank) % B (collisions: last wins), write_pos += kept.
 * key_wrap32 = 1 evaluates node * (max_t + 1) in int32 like the reference does
 * (recency.py:347; it wraps at dataset scale and the reference's results depend on
 * it); key_wrap32 = 0 uses int64 (the intended per-node chronological order).
 * edge_x may be NULL (rows of zeros, recency.py:325-328).  eid0 = store index of
 * the batch's first edge or -1 (recorded in the slot, informational).
 * scratch: 256-byte aligned, >= tgmx_ring_update_scratch_bytes(n, directed) bytes; its first 4096 bytes must be ZERO
 * when the buffer is first handed to the library (a self-resetting barrier of tgmx_recency_step's update workgroups
 * lives there; the library leaves it zero) -- everything behind them is plain scratch.  */
size_t tgmx_ring_update_scratch_bytes(int64_t n, int32_t directed);
int tgmx_ring_update(tgmx_adj_t* ring, int32_t* write_pos, float* ring_x, int32_t D, int32_t B,
                     int32_t num_nodes, const int32_t* src, const int32_t* dst, const int64_t* ts,
                     const float* edge_x, int64_t n, int64_t eid0, int32_t directed, int32_t key_wrap32,
                     int32_t* scratch, int32_t* status, tgmx_stream_t stream);

/* One whole hook call, RecencyNeighborHook.__call__ (recency.py:119-171), as ONE entry point: concatenate the
 * hop-0 seed groups (recency.py:173-237), run every hop's lookup (hop h+1 reads hop h's outputs in place, pads
 * included, recency.py:140-159), then append the batch to the rings (recency.py:161-163 -> :323-399).
 * Semantics are exactly tgmx_ring_lookup x n_hops followed by tgmx_ring_update (or, with indptr set,
 * tgmx_recency_lookup_csr x n_hops against the static index); the seed concatenation rides
 * on the hop-0 lookup launch (its wave of seed s reads group g's arrays and publishes seed_nid0[s] / seed_ts0[s]).
 * Scheduling (results identical): for batches of up to 4096 ring entries (2 per edge, 1 if directed) the update's
 * sort -- and up to 1024 entries also its placement decisions, which read write_pos but write only the scratch -- run
 * in extra workgroups of the hop-0 / hop-1 lookup launches; the rings themselves are written by the launch(es) that
 * follow the last lookup.  Hop 0 and hop 1 are one launch when tgmx_recency_step_plan says so (hop 1's waves re-derive
 * their seed from the unchanged rings instead of waiting for hop 0's output).
 *   timed_hop = 0 or 1 then times that one launch.
 *
 * Riders and workgroup dispatch order (what the in-kernel cross-workgroup waits rely on).  The update workgroups that ride a
 * lookup launch ("riders": one per 256-entry chunk of the batch, at most 16) meet at a barrier inside that launch: two
 * words at the head of `scratch` (arrivals, generation; self-resetting, hence "zero at first use").  A barrier between
 * workgroups of one launch is only safe if all of them are resident at the same time.  The riders are therefore the launch's
 * FIRST workgroups (blockIdx 0 .. riders - 1), and the library relies on the hardware dispatching the workgroups of a launch in
 * ascending blockIdx order: the first 16 workgroups of any launch fit the chip together (one per CU is enough), so they are
 * all dispatched before any lookup workgroup behind them can occupy a slot, whatever the grid size (tests/test_sampler_gpu.py
 * runs them in a launch 7 x larger than what the chip holds at once).  Nothing else waits across workgroups by default: the
 * opt-in tail commit (environment TGMX_TAIL=1: the LAST workgroups of a launch wait for every earlier one, counted at scratch
 * words 2.., then write the rings) relies on the same order -- the workgroups a tail waits for all precede it in dispatch order,
 * so they are running or finished by the time it is resident.  If either wait ever fails to complete -- a scratch head that was
 * not zero at first use, or a device that dispatched out of order -- it gives up after about a second of polling, sets
 * TGMX_ST_SCRATCH in `status` and the call's update is skipped or incomplete (the rings of that batch are then unspecified;
 * the lookups' outputs are unaffected): a reported error, never a hang (test_dirty_scratch_head_is_reported_not_hung).
 * TGMX_NO_RIDE=1 runs the same update as launches of its own (no in-kernel wait at all).
 *
 *   n_groups = 0: hop-0 seeds are already in seed_nid0 / seed_ts0 (S0 of them).
 *   n_hops   = 0: update only.          n = 0: lookups only.
 *   timed_hop >= 0: ev_start / ev_stop are recorded around that hop's lookup launch. */
#define TGMX_MAX_SEED_GROUPS 8
#define TGMX_MAX_HOPS 8
typedef struct tgmx_recency_step {
  tgmx_adj_t* ring;            /* streaming: [num_nodes, B] records.   static index: adj[M] records */
  int32_t* write_pos;          /* streaming: [num_nodes].              static index: NULL */
  float* ring_x;               /* streaming: [num_nodes, B, D].        static index: edge_x[E, D] */
  const int64_t* indptr;       /* NULL = streaming rings; else the static index (tgmx_csr_build): no update, n must be 0 */
  int64_t ev_lo, ev_hi;        /* static index: first edge visible in this epoch / first edge of this batch */
  int32_t D, B, num_nodes;
  int32_t n_groups;
  const int32_t* grp_nid[TGMX_MAX_SEED_GROUPS];
  const int64_t* grp_ts[TGMX_MAX_SEED_GROUPS];
  int64_t grp_n[TGMX_MAX_SEED_GROUPS];
  int32_t* seed_nid0;          /* [S0] hop-0 seeds (output when n_groups > 0, input otherwise) */
  int64_t* seed_ts0;           /* [S0] */
  int64_t S0;                  /* used only when n_groups == 0 */
  int32_t n_hops;
  int32_t k[TGMX_MAX_HOPS];
  int32_t* out_nid[TGMX_MAX_HOPS];  /* [S_h, k_h] */
  int64_t* out_ts[TGMX_MAX_HOPS];   /* [S_h, k_h] */
  float* out_x[TGMX_MAX_HOPS];      /* [S_h, k_h, D] */
  const int32_t* src;          /* batch edges (n = 0: no update) */
  const int32_t* dst;
  const int64_t* ts;
  const float* edge_x;         /* [n, D] or NULL */
  int64_t n, eid0;
  int32_t directed, key_wrap32;
  int32_t* scratch;            /* 256-byte aligned, >= tgmx_ring_update_scratch_bytes(n, directed) bytes */
  int32_t* status;
  int32_t timed_hop;           /* -1: none */
  tgmx_event_t ev_start, ev_stop;
  int64_t ts_bound;            /* 0 = unknown; else a promise: every timestamp of every batch satisfies 0 <= t <= ts_bound
                                  (lets the large-batch update sort only the key bits that can be set; a violation is
                                  reported as TGMX_ST_TS_BOUND and the order of that batch is unspecified) */
  /* ABI v2.  Negatives generated in place: RandomNegativeEdgeSamplerHook.__call__ (tgm/hooks/negatives/sampler.py:45-65)
   * folded into the seed fetch.  With neg_out set, seed group `neg_group` takes no id array (grp_nid[neg_group] is
   * ignored): its i-th id is the i-th draw of tgmx_random_negatives(neg_low, neg_high, ., neg_seed, neg_call) -- the
   * same counter-based generator -- its times are grp_ts[neg_group], and the hop-0 lookup also writes the group to
   * neg_out[grp_n] / neg_time_out[grp_n] (the hook's `neg` / `neg_time`).  neg_out = NULL: off (v1 behaviour). */
  int32_t neg_group, neg_low, neg_high;
  uint64_t neg_seed, neg_call;
  int32_t* neg_out;
  int64_t* neg_time_out;
  /* != 0: if the lookups of this call flag a bad hop-0 seed (TGMX_ST_SEED_RANGE / _TIME), the update of the same call leaves
   * rings and write_pos untouched -- the reference validates the seeds before it changes anything (recency.py:173-237), so a
   * caller that raises on the status word after the call (validate='sync') sees unchanged state, with ONE read-back. */
  int32_t guard_seed_errors;
  /* != 0: a promise that the batch's timestamps are non-decreasing (every batch of a chronological loader over the
   * time-sorted store): the large-batch update then takes max(ts) from the last edge instead of a reduction launch.  Checked on
   * the device: a violation is reported as TGMX_ST_TS_BOUND. */
  int32_t sorted_ts;
  /* Optional, per hop: DELTA writes of the feature rows into PERSISTENT output buffers (a loader's output pool).  A row's
   * neighbors sit at the right end of its k slots and everything left of the leftmost non-pad slot is zero, so a caller
   * that hands the same out_x[h] to consecutive calls -- unmodified in between -- and keeps out_valid[h][row] = the row's
   * SPAN (slots from its leftmost non-pad one to the end; 0 = all pads) needs writes only from slot k - max(previous span,
   * new span) on; the call updates out_valid.  Start with out_x[h] zeroed and out_valid[h] zeroed.  ~60 % of all slots
   * are pads at steady state: this halves the bytes the lookup launch writes.
   * NULL: every slot of every row is written (fresh buffers).  Ids and times are always written in full. */
  int32_t* out_valid[TGMX_MAX_HOPS];
  /* optional with out_valid: receives the span each row held BEFORE the call (the byte accounting of a timed launch needs both) */
  int32_t* out_valid_prev[TGMX_MAX_HOPS];
  /* ABI v4.  Generated negatives of a SHARD: draw i of this call is draw neg_index0 + i of the batch, so that the shares of a
   * batch drawn by different ranks are slices of the ids a single rank would draw for the whole batch (rank order concatenation
   * of the ranks' outputs then equals the single-rank tensors).  0 for an unsharded batch. */
  int64_t neg_index0;
  /* ABI v4, static index only, optional (NULL: off): int64[num_nodes], zero-initialised by the caller and owned by it.  The lookups
   * keep, per node, where its visible prefix ended the last time it was looked up, and start the next batch-boundary search there
   * (a node gains a handful of entries per batch: two probes instead of a search over its whole history).  A cache: results never
   * depend on its contents; any values are safe. */
  int64_t* csr_cursor;
  /* ABI v4, static index only.  != 0: `ring_x` holds the feature rows in ADJACENCY order -- row p belongs to adjacency record p
   * (a copy of edge_x[adj[p].eid] made once at index build) -- so a node's window of B records gathers B consecutive rows instead
   * of B rows scattered by edge id.  0: `ring_x` is edge_x[E, D], addressed by eid. */
  int32_t csr_x_by_pos;
  /* ABI v4, optional per hop: [S_h, k_h] int32, the EDGE ID behind every output slot (-1 for a pad slot; -1 everywhere for batches
   * whose edges carry no store ids, eid0 < 0).  With out_eid[h] set, out_x[h] may be NULL: the feature rows are then not copied at all
   * -- a consumer that holds the resident store (tgmx_tgat_forward: tgmx_tgat_hop_t.nbr_eid) gathers them by id where it uses them,
   * which removes the copy's writes and the consumer's re-read of them (177 of the 244 MB of a wiki-shaped batch). */
  int32_t* out_eid[TGMX_MAX_HOPS];
} tgmx_recency_step_t;

int tgmx_recency_step(const tgmx_recency_step_t* step, tgmx_stream_t stream);
/* How tgmx_recency_step would schedule this argument block: bit 0 = hop 0 and hop 1 run as ONE launch (B <= 64, hop 1
 * not served by the narrow-row kernel; static index: wide rows only).  Informational (bench.py
 * attributes the timed launch's bytes with it); results never depend on it. */
int tgmx_recency_step_plan(const tgmx_recency_step_t* step);

/* ------------------------------------------------------------------------
 * The loader's per-batch call as ONE entry point: DGDataLoader.__call__ (tgm/data/loader.py:158-170: slice,
 * materialize) -> HookManager.execute_active_hooks (tgm/hooks/hook_manager.py:139-168) for the hook chain
 *   [EdgeShardHook (ours: a rank's contiguous share of the batch)] -> RandomNegativeEdgeSamplerHook
 *   (tgm/hooks/negatives/sampler.py:45-65) -> RecencyNeighborHook (tgm/hooks/neighbors/recency.py:119-171)
 * over the device-resident stream.  A batch is the edge range [edge_lo, edge_lo + n_edges) of the store; the seeds are
 * the roles listed in seed_role (in the order of the hook's seed_nodes_keys), every role over this rank's share
 * [n * rank / world, n * (rank + 1) / world) of the batch, all with the share's edge times as query times;
 * negatives are generated inside the seed fetch (tgmx_recency_step_t.neg_out).  Streaming mode (step.indptr == NULL,
 * update != 0) appends the WHOLE batch to the rings after the lookups (every rank replays the global batch).
 * The caller owns the outputs (a pool of preallocated buffers): nothing is allocated, nothing synchronises.
 * Results are identical to running the three hooks one by one. */
#define TGMX_SEED_SRC 0
#define TGMX_SEED_DST 1
#define TGMX_SEED_NEG 2
typedef struct tgmx_pipeline {
  const int32_t* src;          /* resident stream, time-sorted (DGData, tgm/data/dg_data.py:350-394) */
  const int32_t* dst;
  const int64_t* ts;
  const float* edge_x;         /* [num_edges, D] or NULL */
  int64_t num_edges;
  int32_t rank, world;         /* seeds come from this rank's share of every batch (world = 1: the whole batch) */
  int32_t n_roles;
  int32_t seed_role[TGMX_MAX_SEED_GROUPS]; /* TGMX_SEED_* per seed group */
  int32_t neg_low, neg_high;   /* negatives: uniform ids in [neg_low, neg_high) */
  uint64_t neg_seed;
  int32_t update;              /* streaming mode: append the batch to the rings after the lookups */
  int32_t reserved0;
  tgmx_recency_step_t step;    /* static fields filled by the caller: state pointers, D, B, num_nodes, n_hops, k[],
                                  directed, key_wrap32, scratch, status, ts_bound; static index: indptr, ev_lo
                                  (first edge visible in this epoch; ev_hi is set to edge_lo by the call) */
} tgmx_pipeline_t;

typedef struct tgmx_pipeline_out {
  int32_t* neg;                /* [share]  (NULL unless a TGMX_SEED_NEG role exists) */
  int64_t* neg_time;           /* [share] */
  int32_t* seed_nid0;          /* [n_roles * share] concatenated hop-0 seeds */
  int64_t* seed_ts0;
  int32_t* out_nid[TGMX_MAX_HOPS];
  int64_t* out_ts[TGMX_MAX_HOPS];
  float* out_x[TGMX_MAX_HOPS];
  int32_t timed_hop;           /* -1: none; else ev_start / ev_stop bracket that hop's lookup launch */
  tgmx_event_t ev_start, ev_stop;
  int32_t* out_valid[TGMX_MAX_HOPS]; /* optional: tgmx_recency_step_t.out_valid of this output set (delta feature writes) */
  int32_t* out_valid_prev[TGMX_MAX_HOPS]; /* optional: tgmx_recency_step_t.out_valid_prev */
  int32_t* out_eid[TGMX_MAX_HOPS];   /* optional: tgmx_recency_step_t.out_eid (out_x[h] may then be NULL) */
} tgmx_pipeline_out_t;

/* Optional tail of the chain, for the TGN loop: DeduplicationHook (tgm/hooks/dedup.py:35-67) over [batch src | batch dst |
 * negatives | every hop's neighbor ids] and the sampled edge list of one hop (tgmx_tgn_edge_list), enqueued behind the sampler
 * in the same call; the three sizes the host must learn (unique count, dedup status, edge count) are copied to pinned host
 * memory asynchronously and `sizes_ready` is recorded behind the copy. */
typedef struct tgmx_pipeline_post {
  int32_t dedup;               /* 1: unique ids */
  int32_t dedup_neg, dedup_nbr;/* include the negatives / the sampled neighbor ids */
  int32_t num_nodes;
  void* dedup_ws;              /* tgmx_unique_ids workspace (zero at first use) */
  int32_t* uniq_out;           /* [min(total ids, num_nodes)] */
  int32_t edge_hop;            /* -1: no edge list */
  int64_t edge_cap;            /* >= rows * k of that hop */
  int64_t* row_off;            /* [rows + 1] scratch */
  int64_t* edge_index;         /* [2, edge_cap] */
  int64_t* edge_t;             /* [edge_cap] */
  float* edge_x;               /* [edge_cap, D] */
  int64_t* dev_sizes;          /* device [3]: unique count | status (low word) | edge count; status word zero at first use */
  int64_t* host_sizes;         /* pinned host [3], device-accessible (hipHostMalloc / torch pin_memory): written by a kernel's stores */
  tgmx_event_t sizes_ready;
} tgmx_pipeline_post_t;

/* ABI v7 -- a LAUNCH WORKER: one host thread of the library that issues tgmx_pipeline_step for the caller, in submission order, on the
 * stream the caller names (the loader's side stream), bracketed by `wait_ev` (NULL or: the stream first waits for it) and `record_ev`
 * (NULL or: recorded behind the step's last launch).  The argument blocks are COPIED at submission.  tgmx_worker_wait blocks until
 * the job with that ticket has been issued (not until the device has run it) and returns ITS status (tgmx_last_error of the calling
 * thread then holds its text).  A host-side convenience with no counterpart in the reference (its loader runs in the caller's
 * thread, tgm/data/loader.py:158-170): DGDataLoader(side_stream=True) uses it so that the consumer's thread does not pay for the
 * producer's launches.  The worker issues on the device that was current when it was created. */
typedef void* tgmx_worker_t;
int tgmx_worker_create(tgmx_worker_t* w);
int tgmx_worker_destroy(tgmx_worker_t w); /* finishes the pending jobs, joins the thread */
/* Byte accounting of one lookup launch (bench / profiling), as TGMX_ACCOUNTING_PARTIALS partial triples that the caller adds
 * up whenever it likes (one launch, nothing to zero): column 0 = non-pad slots of ids[0 .. slots), column 1 = sum over rows of
 * max(span_prev, span_cur) (slots whose feature row was rewritten), column 2 = sum of span_cur.  counts: device,
 * int64[TGMX_ACCOUNTING_PARTIALS][3], overwritten.  span_* may be NULL (columns 1 and 2 are then zero). */
#define TGMX_ACCOUNTING_PARTIALS 1024
int tgmx_lookup_accounting(const int32_t* ids, int64_t slots, const int32_t* span_prev, const int32_t* span_cur, int64_t rows,
                           int64_t* counts, tgmx_stream_t stream);
int tgmx_worker_pipeline_step(tgmx_worker_t w, const tgmx_pipeline_t* pipe, int64_t edge_lo, int64_t n_edges, uint64_t neg_call,
                              const tgmx_pipeline_out_t* out, const tgmx_pipeline_post_t* post /* NULL: none */, tgmx_stream_t stream,
                              tgmx_event_t wait_ev, tgmx_event_t record_ev, uint64_t* ticket);
int tgmx_worker_wait(tgmx_worker_t w, uint64_t ticket);
int tgmx_pipeline_step(const tgmx_pipeline_t* pipe, int64_t edge_lo, int64_t n_edges, uint64_t neg_call,
                       const tgmx_pipeline_out_t* out, const tgmx_pipeline_post_t* post /* NULL: none */, tgmx_stream_t stream);

/* DGStorageArrayBackend._binary_search (tgm/core/_storage/backends/array_backend.py:301-321): event index range
 * [*lb, *ub) of the slice {start_time <= t <= end_time (inclusive; has_* = 0: unbounded), start_idx <= i < end_idx
 * (negative: unbounded)} over the time-sorted timeline.  `host_time` is a HOST array (the one exception to the
 * device-pointer convention: slicing is host arithmetic, O(log n), no device work). */
int tgmx_slice(const int64_t* host_time, int64_t n, int32_t has_start_time, int64_t start_time, int32_t has_end_time,
               int64_t end_time, int64_t start_idx, int64_t end_idx, int64_t* lb, int64_t* ub);

/* ring.fill(pad), write_pos.zero_()  (recency.py:111-117) */
int tgmx_ring_reset(tgmx_adj_t* ring, int32_t* write_pos, int32_t B, int32_t num_nodes,
                    tgmx_stream_t stream);

/* ------------------------------------------------------------------------
 * Static index build (ours: the reference has no index; SURVEY.md Appendix A.3 shows the ring state at the start of
 * a batch is a pure function of it).  Orders node n's adjacency entries by (batch_idx, time, role, eid) -- role 0 =
 * n is the source -- with one device radix sort and writes indptr[num_nodes + 1] and the 16-byte records.
 *   src/dst/ts   the time-sorted stream (the store DGData normalises, tgm/data/dg_data.py:350-394)
 *   batch_starts [num_batches] increasing first-edge indices of the loader's batches (tgm/data/loader.py:158-170);
 *                edges before batch_starts[0] form one leading batch; num_batches = -1: every edge is its own batch,
 *                i.e. plain (eid, role) order -- the candidate order of the uniform sampler (array_backend.py:127-135)
 *   directed     index source-role entries only (recency.py:332-343)
 *   workspace    >= tgmx_csr_build_workspace_bytes(num_edges, num_nodes, directed)
 *   status       TGMX_ST_EDGE_RANGE is raised for endpoints outside [0, num_nodes)
 * ------------------------------------------------------------------------ */
size_t tgmx_csr_build_workspace_bytes(int64_t num_edges, int32_t num_nodes, int32_t directed);
int tgmx_csr_build(const int32_t* src, const int32_t* dst, const int64_t* ts, int64_t num_edges, int32_t num_nodes,
                   const int64_t* batch_starts, int64_t num_batches, int32_t directed, int64_t* indptr,
                   tgmx_adj_t* adj, void* workspace, size_t workspace_bytes, int32_t* status, tgmx_stream_t stream);

/* ------------------------------------------------------------------------
 * TGAT / TemporalAttention forward (fp32, eval mode).  The attention is
 * restructured (q-length 1 => the W_KV projection folds onto the query / output
 * side, see csrc/tgat.hip); these are its building blocks, composed by
 * tgm_amd/nn/tgat.py in the order of tgm/nn/encoder/tgat.py:95-149.
 * ------------------------------------------------------------------------ */

/* nn.Dropout(p) in training mode as a counter-based mask (attention.py:119,126; PyG TransformerConv's dropout on the
 * attention weights): element i of the tensor a site covers is kept iff the 32 random bits of (seed, stream, row0 * width
 * + i) are >= floor(p * 2^32), and scaled by 1 / (1 - p).  Nothing is stored: the backward entry points regenerate the
 * mask from the same descriptor.  p = 0 (or a NULL descriptor): dropout off. */
typedef struct tgmx_dropout {
  float p;
  uint64_t seed;
  uint64_t stream; /* one value per (forward call, site) */
  int64_t row0;    /* index of the first row this call covers within the site's tensor (level-by-level calls) */
} tgmx_dropout_t;

/* out[r, c] = x[r, c] * mask(r, c) / (1 - p), c < C  (in place when out == x; forward on a tensor, backward on its gradient) */
int tgmx_dropout(const float* x, int64_t ldx, int64_t R, int32_t C, const tgmx_dropout_t* drop, float* out, int64_t ldo,
                 tgmx_stream_t stream);

/* out[n, T] = cos(fma(float(x[i]), w[t], b[t]))        Time2Vec.forward
 * (tgm/nn/modules/time_encoding.py:22-24); x is int64 (x_is_int64) or float32. */
int tgmx_time2vec(const void* x, int32_t x_is_int64, const float* w, const float* b, int32_t T,
                  int64_t n, float* out, tgmx_stream_t stream);

/* out[i, :] = table[idx[i], :], negative idx wraps (pad id -1 -> last row)
 * (tgm/nn/encoder/tgat.py:128-130 leaf features). */
int tgmx_gather_rows(const float* table, int64_t num_rows, int32_t dim, const int32_t* idx,
                     int64_t n, float* out, int64_t ldo, tgmx_stream_t stream);

/* out[R, O] = [x[:, :d] | 0 pad | time_feat]: residual / query input
 * (tgm/nn/modules/attention.py:93-95).  time_feat [R, T] may be NULL: then it is
 * Time2Vec(0) = cos(tb), what TGAT.forward passes (tgat.py:139). */
int tgmx_tgat_rres(const float* x, int64_t ldx, int32_t d, const float* tb, const float* time_feat,
                   int32_t T, int32_t O, int64_t R, float* out, int64_t ldo /* 0 = O; columns [O, ldo) zeroed */,
                   tgmx_stream_t stream);

/* C[b] = act(A[b] (M x K, lda) * B[b]^T (B is N x K, ldb) + bias[b]), b < batch with
 * element strides (bias: [batch, N], i.e. [N] when batch = 1); exact-fp32 MFMA.  Replaces the nn.Linear calls of
 * attention.py:96,125 and tgat.py:36-38 and the folded W_K / W_V contractions.
 * Determinism: a call is reproducible bit for bit.  The ORDER in which a row's K products are summed depends on the kernel the call
 * picks from M alone -- M <= 2048: eight contiguous K slices per output block; M >= 6144:
 * one ascending chain through 32-wide K chunks staged in LDS; otherwise four interleaved k-step sets -- so one
 * row computed in a 600-row call and in a 12 600-row call can differ in the last bits (all three are pinned against float64 in
 * tests/test_gemm_gpu.py).  Every bit-identity claim of this library (compact rows, sharded vs single-rank outputs) is between
 * calls whose M falls in the same band per GEMM; a rank whose share of a batch lands in another band is outside it --
 * TGMX_GEMM_SMALL=0 and TGMX_GEMM_LDS=0 pin the four-set kernel for every M when that matters more than the time (the 600-row
 * layer's 8 us; 15-30 % of a many-row GEMM). */
int tgmx_sgemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                  int64_t M, int32_t N, int32_t K, const float* bias, int32_t relu, int32_t batch,
                  int64_t strideA, int64_t strideB, int64_t strideC, tgmx_stream_t stream);

/* Per-row masked softmax attention over the k sampled slots
 * (attention.py:103-122 after folding): qf [R,H,C], nbrf [R,k,d], ex [R,k,D]
 * -> zbar [R,H,C], C = d + D + T.  Neighbor time features are either computed in
 * the kernel, cos(fma(float(seed_t[r] - nbr_t[r,s]), tw, tb)) (tgat.py:143-145), or
 * read from nbr_time_feat [R,k,T] when it is non-NULL; the valid-neighbor mask is
 * nbr_id[r,s] != -1, or mask [R,k] (bytes) when it is non-NULL. */
int tgmx_tgat_attn_reduce(const float* qf, const float* nbrf, int32_t d, const float* ex, int32_t D,
                          const int64_t* seed_t, const int64_t* nbr_t, const int32_t* nbr_id,
                          const float* tw, const float* tb, const float* nbr_time_feat,
                          const uint8_t* mask, int32_t T, int32_t H, int32_t k, int64_t R,
                          float scale, int32_t head_stride /* floats between heads in qf / zbar rows, 0 = C */,
                          float* zbar, float* attn_probs /* optional [R,H,k]: softmax weights BEFORE dropout, for backward */,
                          const tgmx_dropout_t* drop /* NULL = off; element (r, h, s) of [R, H, k], attention.py:119 */,
                          tgmx_stream_t stream);

/* out[R, O + d0] = [LayerNorm(y + res) * gamma + beta | z0]
 * (attention.py:127 + the concat of tgat.py:36). */
int tgmx_ln_residual_concat(const float* y, int64_t ldy, const float* res, int64_t ldr,
                            const float* gamma, const float* beta, int32_t O, float eps, const float* z0,
                            int32_t d0, int64_t R, float* out, int64_t ldo /* leading dims: 0 = dense */,
                            tgmx_stream_t stream);

/* Whole TGAT forward in one call (tgm/nn/encoder/tgat.py:95-149): enqueues the leaf gathers and,
 * per layer, the row-batched kernel sequence above.  hops[i] are the sampler's hop-i outputs
 * (rows_i = S0 * k_0 * ... * k_{i-1}); workspace from tgmx_tgat_workspace_bytes(); out [S0, emb]. */
#define TGMX_TGAT_MAX_LAYERS 4
/* Weight matrices are passed as zero-padded copies whose rows start 16-byte aligned
 * (p4(x) = x rounded up to a multiple of 4; dh = O / H; C = d + D + T): */
typedef struct tgmx_tgat_layer {
  const float* W_Q;   /* [O, p4(O)]          attn.{l}.W_Q.weight, rows padded                          */
  const float* W_K_t; /* [C, H * p4(dh)]     W_KV[:O]^T, head h's dh columns at column h * p4(dh)      */
  const float* W_V;   /* [O, p4(C)]          W_KV[O:], rows padded                                     */
  const float* W_O;   /* [O, p4(O)]                                                                    */
  const float* b_O;   /* [O]                                                                           */
  const float* ln_g;  /* [O]  layer_norm.weight                                                        */
  const float* ln_b;  /* [O]                                                                           */
  const float* fc1_w; /* [emb, p4(O + d0)]   merge_layers.{l}.fc1.weight, rows padded                  */
  const float* fc1_b; /* [emb]                                                                         */
  const float* fc2_w; /* [emb_out, p4(emb)]                                                            */
  const float* fc2_b; /* [emb_out]                                                                     */
  /* optional (inference): the query side folded onto the layer input -- qf[r, h, c] = qf_v[h*p4(C) + c] +
   * sum_j x[r, j] * qf_U[(h*p4(C) + c) * p4(d) + j], with U_h = W_K,h^T W_Q,h[:, :d] and v_h = W_K,h^T W_Q,h[:, O-T:]
   * cos(tb) (the residual's time part is Time2Vec(0), attention.py:93-95).  NULL: Q and qf are computed per call. */
  const float* qf_U;  /* [H * p4(C), p4(d)] */
  const float* qf_v;  /* [H * p4(C)]        */
  /* optional (inference): the four weights behind the attention reduce in the 16 x 16-tiled layout of tgmx_tgat_tile16()
   * (made once per parameter version).  With all four set, everything behind the attention reduce of a layer -- W_V, W_O,
   * LayerNorm(y + residual), concat, fc1 + ReLU, fc2 -- runs as ONE kernel over 16-row tiles.  NULL: the row-major copies. */
  const float* W_V_t16; /* H blocks, head h = tile16(W_KV[O + h*dh : O + (h+1)*dh], dh, C) at h * tile16_floats(dh, C) */
  const float* W_O_t16; /* tile16(W_O, O, O)                */
  const float* fc1_t16; /* tile16(fc1.weight, emb, O + d0)  */
  const float* fc2_t16; /* tile16(fc2.weight, emb_out, emb) */
  /* optional (inference, d == 1 -- a model whose nodes carry ONE feature, like the reference's example configuration): qf_U / qf_v
   * in the order the register-resident attention kernel's lanes consume them, so that the kernel evaluates qf = qf_v + x * qf_U
   * itself and the [rows, H * p4(C)] qf buffer is neither written nor read.  [H, 64, 16] floats; entry (h, lane):
   *   0..3  qf_v of the edge columns d + 4 lane + i      4..7  qf_U of the same columns
   *   8, 9  qf_v, qf_U of time column d + D + lane       10, 11  the same for time column d + D + lane + 64
   *   12, 13  qf_v, qf_U of neighbor column `lane`       14, 15  zero;   columns that do not exist: zero.       NULL: qf is a buffer. */
  const float* qf_lane;
  /* optional (inference, H == 2 and 2 * ceil(dh / 16) <= 12): both heads' W_V for the one-kernel tail's stage 1 as ONE tiled matrix --
   * tile16(S, 2 * 16 * ceil(dh / 16), C) where S stacks the heads, head h's dh rows at row h * 16 * ceil(dh / 16), zero rows between:
   * the stage then runs both heads' output blocks in one pass.  NULL: the per-head images of W_V_t16. */
  const float* W_V_t16c;
  int32_t d, D, T, O, H, emb, emb_out;
  float ln_eps;
} tgmx_tgat_layer_t;
typedef struct tgmx_tgat_model {
  const float* tw; /* [T] time_encoder.w.weight */
  const float* tb; /* [T] time_encoder.w.bias   */
  int32_t num_layers, d0;
  tgmx_tgat_layer_t layers[TGMX_TGAT_MAX_LAYERS];
  /* training (save != 0) with dropout: p and seed; `stream` = the forward call's counter -- layer j (1-based) uses
   * stream * 64 + 2 j for the attention weights and stream * 64 + 2 j + 1 for the W_O output (attention.py:119,126) */
  tgmx_dropout_t drop;
} tgmx_tgat_model_t;
typedef struct tgmx_tgat_hop {
  const int64_t* seed_t; /* [rows_i]        seed_times[i]    */
  const int32_t* nbr_id; /* [rows_i, k]     nbr_nids[i]      */
  const int64_t* nbr_t;  /* [rows_i, k]     nbr_edge_time[i] */
  const float* edge_x;   /* [rows_i, k, D]  nbr_edge_x[i]; NULL with nbr_eid set */
  int32_t k;
  /* ABI v5 (the 4 bytes of padding behind k in v4).  != 0 on hop i >= 1: a PROMISE that row r of this hop is a function of its seed
   * (hops[i-1].nbr_id[r], hops[i-1].nbr_t[r]) alone -- true when all hops come from one sampler call (hop i is seeded with hop i-1's
   * flattened outputs and no lookup of a call changes the sampler's state: tgm/hooks/neighbors/recency.py:141-143, 161-163).  Slots
   * with equal (id, time) are then the same row of every deeper level, and inference (save == 0) with every hop >= 1 so marked
   * computes each distinct row ONCE (tgmx_pair_dedup) and lets the level above read it by index -- the embeddings are the ones the
   * row-per-slot computation gives, bit for bit (a row's arithmetic does not depend on where it sits).  Pads alone -- one pair, (-1, 0)
   * -- are 35-60 % of a level at the headline shape.  0: nothing is assumed (hand-made inputs, the reference's unit tests). */
  int32_t seed_keyed;
  /* ABI v4: edge features by id.  With edge_x == NULL and nbr_eid / edge_table set, slot (r, s) reads edge_table[nbr_eid[r, s]]
   * ([E, D] rows of the resident store; -1: a pad slot, zeros) where the attention consumes it -- the sampler then never writes
   * the dense [rows, k, D] copy (tgmx_recency_step_t.out_eid) and the attention never re-reads it.  Needs the register-resident
   * attention kernel (n_heads <= 2, k <= 20, D a multiple of 4); otherwise TGMX_E_UNSUPPORTED: gather the rows first.  With save != 0
   * too (round 3, later): tgmx_tgat_backward then reads the same rows by id in its attention backward. */
  const int32_t* nbr_eid;  /* [rows_i, k] */
  const float* edge_table; /* [E, D] */
} tgmx_tgat_hop_t;
/* Where tgmx_tgat_forward keeps its intermediates (offsets in floats from the 256-byte aligned
 * workspace base; -1 = not kept).  Row strides: rres/oattn/y Op, Q H*dhp, qf/zbar H*Cp, cat Kc, h1 Ep,
 * probs H*k, out emb_out.  Level i of the hop tree occupies rows [level_off[i], level_off[i+1]). */
typedef struct tgmx_tgat_layer_layout {
  int64_t R, rres, oattn, y, Q, qf, zbar, cat, h1, probs, out;
  int32_t Op, dhp, Cp, Kc, Ep;
} tgmx_tgat_layer_layout_t;
typedef struct tgmx_tgat_layout {
  int64_t total_bytes, z0;
  int64_t level_rows[TGMX_TGAT_MAX_LAYERS + 1], level_off[TGMX_TGAT_MAX_LAYERS + 2];
  tgmx_tgat_layer_layout_t layers[TGMX_TGAT_MAX_LAYERS];
  /* ABI v5, compact rows (tgmx_tgat_hop_t.seed_keyed; -1 = level not deduplicated).  Level i's block, n = level_rows[i] int32 each:
   * [count + 15 pad | uniq n | owner n | cidx n | rep n | tgmx_pair_dedup workspace]: rep[r] = the compact row of original row r,
   * uniq[c] = the original row compact row c stands for, count = the number of compact rows (all three device-side). */
  int64_t compact[TGMX_TGAT_MAX_LAYERS + 1];
} tgmx_tgat_layout_t;
/* Weight [N, K] (row stride ldw floats) -> out[tgmx_tgat_tile16_floats(N, K)]: block (nb, kb) of 16 x 16 is 256 consecutive
 * floats, element 4 * lane + j = W[16 nb + (lane & 15)][16 kb + 4 (lane >> 4) + j], zero outside the matrix -- the order the
 * 16x16x4 fp32 MFMA's A operand is fetched in, so one wave load is 1 KB of consecutive memory. */
/* Batched 2-D repacking of small matrices -- a module's weights into the zero-padded / transposed layouts the kernels read -- as ONE
 * launch (round 3: the training step rebuilt them after every optimizer step with ~25 torch launches).  Job i writes, for r < dst_rows and
 * c < dst_cols, dst[r * dst_ld + c] = src[r * src_ld + c] (transpose == 0) or src[c * src_ld + r] (transpose != 0) where r < rows and
 * c < cols (the extent of the data in dst coordinates), 0 elsewhere.  Replaces torch.zeros + slice assignment (tgm_amd/nn/tgat.py). */
#define TGMX_PACK_MAX_JOBS 32
typedef struct tgmx_pack_job {
  const float* src;
  float* dst;
  int64_t src_ld, dst_ld;
  int32_t rows, cols, dst_rows, dst_cols;
  int32_t transpose, reserved_;
} tgmx_pack_job_t;
int tgmx_pack2d(const tgmx_pack_job_t* jobs, int32_t n_jobs, tgmx_stream_t stream);
size_t tgmx_tgat_tile16_floats(int32_t N, int32_t K);
int tgmx_tgat_tile16(const float* W, int64_t ldw, int32_t N, int32_t K, float* out, tgmx_stream_t stream);
/* ABI v5.  The distinct (id, time) pairs among n of them (one level of the hop tree; see tgmx_tgat_hop_t.seed_keyed): owner[r] = the
 * smallest index carrying r's pair, *count = number of distinct pairs, numbered densely: cidx[owner] = the pair's number (written for
 * owners only), uniq[number] = owner.  The numbering order is unspecified.  An open-addressing table in `workspace` (4-byte aligned,
 * tgmx_pair_dedup_workspace_bytes(n), contents irrelevant on entry); a memset and two launches.  n <= 2^28. */
size_t tgmx_pair_dedup_workspace_bytes(int64_t n);
int tgmx_pair_dedup(const int32_t* ids, const int64_t* times, int64_t n, int32_t* uniq, int32_t* owner, int32_t* cidx,
                    int32_t* count, void* workspace, size_t workspace_bytes, tgmx_stream_t stream);
int tgmx_tgat_layout(const tgmx_tgat_model_t* model, int64_t S0, const tgmx_tgat_hop_t* hops,
                     int32_t save, tgmx_tgat_layout_t* out);
size_t tgmx_tgat_workspace_bytes(const tgmx_tgat_model_t* model, int64_t S0, const tgmx_tgat_hop_t* hops);
/* save != 0: every layer keeps its intermediates + attention weights (training); the workspace then
 * has to be tgmx_tgat_layout(..., save=1).total_bytes large and must outlive the backward pass. */
int tgmx_tgat_forward(const tgmx_tgat_model_t* model, const float* node_x, int64_t num_nodes,
                      const int32_t* seed_ids, int64_t S0, const tgmx_tgat_hop_t* hops,
                      float* workspace, size_t workspace_bytes, int32_t save, float* out,
                      tgmx_stream_t stream);

/* ---- TGAT backward building blocks (training; composed by tgm_amd/nn/tgat.py) ---------------- */

/* C[b] (+)= A[b]^T B[b]: C[m, n] = sum_r A[r, m] * B[r, n] (weight gradients; reduction over rows),
 * exact-fp32 MFMA, deterministic two-stage reduction.  workspace: tgmx_sgemm_tn_workspace_bytes(). */
size_t tgmx_sgemm_tn_workspace_bytes(int64_t R, int32_t M, int32_t N, int32_t batch);
int tgmx_sgemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                  int64_t R, int32_t M, int32_t N, int32_t batch, int64_t strideA, int64_t strideB,
                  int64_t strideC, int32_t accumulate, float* workspace, tgmx_stream_t stream);

/* out[c] (+)= sum_r in[r * ld + c], c < C (bias / LayerNorm / Time2Vec gradients); workspace 256*C floats */
int tgmx_colsum(const float* in, int64_t ld, int64_t R, int32_t C, float* out, int32_t accumulate,
                float* workspace, tgmx_stream_t stream);

/* grad[r, c] = act[r, c] > 0 ? grad[r, c] : 0 */
int tgmx_relu_mask(float* grad, int64_t ldg, const float* act, int64_t lda, int64_t R, int32_t C,
                   tgmx_stream_t stream);

/* dst[r, c] (+)= src[r, c], c < C */
int tgmx_add_cols(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t R, int32_t C,
                  int32_t accumulate, tgmx_stream_t stream);

/* LayerNorm(y + res) backward: du [R, O] (gradient of y and of res), dgx = dout * xhat (its column sum
 * is d gamma; d beta is the column sum of dout). */
int tgmx_ln_backward(const float* dout, int64_t ldd, const float* y, int64_t ldy, const float* res,
                     int64_t ldr, const float* gamma, int32_t O, float eps, int64_t R, float* du,
                     int64_t ldu, float* dgx, int64_t ldg, tgmx_stream_t stream);

/* Backward of tgmx_tgat_attn_reduce (needs the saved attention weights): given dzbar [R,H,Cs] ->
 * dqf [R,H,Cs], dnbr [R,k,d] (accumulated, may be NULL) and per-row partials dtime_rows [R, 2T] of the
 * Time2Vec weight | bias gradients (column-sum them).  k * H <= 64. */
int tgmx_tgat_attn_backward(const float* qf, const float* probs, const float* dzbar, const float* nbrf,
                            int32_t d, const float* ex, int32_t D, const int64_t* seed_t,
                            const int64_t* nbr_t, const float* tw, const float* tb, int32_t T, int32_t H,
                            int32_t k, int64_t R, float scale, int32_t head_stride, float* dqf,
                            float* dnbr, float* dtime_rows, const tgmx_dropout_t* drop /* the forward's, NULL = off */,
                            tgmx_stream_t stream);

/* ---- The whole TGAT backward as ONE call (round 3) ------------------------------------------------------------------------
 * What tgm_amd/nn/_tgat_train.py composed from the building blocks above, launch by launch from Python (~100 ctypes calls and
 * ~60 torch launches per step: the training step was host-bound), driven from C++ in the same order with the same kernels -- the
 * gradients are bit-identical to the composed path (TGMX_TGAT_BWD=py keeps it for the tests), d tb to one rounding (its
 * -sin(tb) * colsum term is evaluated by our kernel instead of torch's).  `saved` is the workspace of the
 * tgmx_tgat_forward(..., save = 1) call with the same model / layout / hops; `dz` [S0, emb_out of the last layer] (row stride ldz);
 * `drop` the forward's dropout block (model->drop at that call; NULL or p = 0: off).  Every gradient has its parameter's own
 * shape, contiguous (W_KV: [2 O, C], keys then values; tw: [T]).  workspace: tgmx_tgat_backward_workspace_bytes(). */
typedef struct tgmx_tgat_layer_grads {
  float *W_Q, *W_KV, *W_O, *b_O, *ln_g, *ln_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} tgmx_tgat_layer_grads_t;
typedef struct tgmx_tgat_grads {
  float *tw, *tb;
  tgmx_tgat_layer_grads_t layers[TGMX_TGAT_MAX_LAYERS];
} tgmx_tgat_grads_t;
size_t tgmx_tgat_backward_workspace_bytes(const tgmx_tgat_model_t* model, const tgmx_tgat_layout_t* layout, const tgmx_tgat_hop_t* hops);
int tgmx_tgat_backward(const tgmx_tgat_model_t* model, const tgmx_tgat_layout_t* layout, const tgmx_tgat_hop_t* hops,
                       const float* saved, const float* dz, int64_t ldz, const tgmx_dropout_t* drop,
                       const tgmx_tgat_grads_t* grads, float* workspace, size_t workspace_bytes, tgmx_stream_t stream);

/* ------------------------------------------------------------------------
 * TGN memory module (tgm/nn/encoder/tgn.py:80-251), forward arithmetic.
 * A node's stored events ("the events of the last batch in which it appeared",
 * per role) are a window (st_lo[node], st_cnt[node]) into an append-only device log
 * {log_other int32, log_t int64, log_raw [.,D] f32}.
 * ------------------------------------------------------------------------ */

/* Replace the store of every node of a batch role (tgn.py:218-229).  Entries are given
 * in node-sorted (stable) order: perm[p] = batch entry at sorted position p, node_sorted[p]
 * its node, [left[p], right[p]) the sorted run of that node; log rows [base, base+n) are written. */
int tgmx_tgn_store(const int64_t* perm, const int32_t* node_sorted, const int64_t* left,
                   const int64_t* right, const int32_t* other, const int64_t* t, const float* raw,
                   int32_t D, int64_t n, int64_t base, int32_t* log_other, int64_t* log_t,
                   float* log_raw, int64_t* st_lo, int32_t* st_cnt, tgmx_stream_t stream);

/* aggr[r] = Last / Mean aggregation of node nodes[r]'s stored messages
 * [mem[v] | mem[other] | raw | Time2Vec(t - last_update[v])], source-role store first
 * (tgn.py:191-209, 43-74); new_lu[r] = max stored t (0 if none); aggr = 0 if none. */
int tgmx_tgn_aggregate(const int32_t* nodes, int64_t R, const float* memory,
                       const int64_t* last_update, int32_t M, int32_t num_nodes,
                       const int64_t* st_lo_s, const int32_t* st_cnt_s, const int64_t* st_lo_d,
                       const int32_t* st_cnt_d, const int32_t* log_other, const int64_t* log_t,
                       const float* log_raw, int32_t D, const float* tw, const float* tb, int32_t T,
                       int32_t mean, float* aggr, int64_t* new_lu,
                       int64_t* assoc /* optional [num_nodes]: assoc[nodes[r]] = (stamp << 32) | r, the reference's
                                         self._assoc[n_id] = arange (tgn.py:193); NULL = not recorded */,
                       int64_t stamp, tgmx_stream_t stream);

/* TGNMemory.update_state in train mode (tgn.py:165-177) when the rows of the preceding forward are at hand: the
 * reference commits _get_updated_memory(unique(src, dst)) -- for an unchanged state exactly the rows that forward produced
 * -- so the commit is memory[v] = val[row(v)], last_update[v] = lu[row(v)] with row(v) from `assoc` (stamp-checked: a
 * node that was not part of that forward sets *status to 1 and is skipped).  Duplicate nodes write identical values. */
int tgmx_tgn_commit_assoc(const int32_t* src, const int32_t* dst, int64_t n, const int64_t* assoc, int64_t stamp,
                          const float* val, const int64_t* lu, int32_t M, int32_t num_nodes, float* memory,
                          int64_t* last_update, int32_t* status, tgmx_stream_t stream);

/* TGNMemory._update_msg_store for both roles of one batch of n <= 1024 events (tgn.py:218-229, called at :173,176) in one
 * launch: log rows [base, base + n) = events grouped by src (stable), [base + n, base + 2n) grouped by dst; every node's
 * (lo, cnt) window is replaced. */
int tgmx_tgn_store_batch(const int32_t* src, const int32_t* dst, const int64_t* t, const float* raw, int32_t D, int32_t n,
                         int64_t base, int32_t* log_other, int64_t* log_t, float* log_raw, int64_t* st_lo_s,
                         int32_t* st_cnt_s, int64_t* st_lo_d, int32_t* st_cnt_d, tgmx_stream_t stream);

/* Compaction of TGNMemory's append-only message log (ours: the reference keeps a Python dict of per-node tensors, tgn.py:218-229): the live
 * windows -- (st_lo, st_cnt) per role and node -- are packed, in (role, node) order, to the front of the fresh log tensors new_*, and
 * st_lo_* are re-pointed (0 for an empty window).  The caller knows the total (the sum of the counts) and sizes new_* for it.
 * workspace >= tgmx_tgn_compact_workspace_bytes(num_nodes).  One scan + one move launch. */
size_t tgmx_tgn_compact_workspace_bytes(int32_t num_nodes);
int tgmx_tgn_compact(int64_t* st_lo_s, const int32_t* st_cnt_s, int64_t* st_lo_d, const int32_t* st_cnt_d, int32_t num_nodes,
                     const int32_t* old_other, const int64_t* old_t, const float* old_raw, int32_t D, int32_t* new_other, int64_t* new_t,
                     float* new_raw, void* workspace, size_t workspace_bytes, tgmx_stream_t stream);

/* TGNMemory._get_updated_memory (tgn.py:191-216) and GraphAttentionEmbedding.forward (tgn.py:30-40, PyG TransformerConv) as
 * ONE call each (inference / no-grad paths): the same launches, in the same order, as the building blocks above composed by
 * the host -- aggregate, gather, the two GRU GEMMs, gates; four stacked projections, edge encoding, lin_edge GEMM, segment
 * sort (or, with tgt_count / cursor / order_big, a counting grouping: histogram inside the edge encoding's launch, one scan, one placement
 * launch, segments sorted by edge id where the attention reads them -- same results), attention.  Every buffer is the caller's.  qkvs is [4, U, H*C]: query | key | value | skip; the result is its 4th
 * block (skip + attention). */
typedef struct tgmx_tgn_memory_fwd {
  const int32_t* nodes; int64_t R;
  const float* memory; const int64_t* last_update; int32_t M, num_nodes;
  const int64_t* st_lo_s; const int32_t* st_cnt_s; const int64_t* st_lo_d; const int32_t* st_cnt_d;
  const int32_t* log_other; const int64_t* log_t; const float* log_raw; int32_t D;
  const float* tw; const float* tb; int32_t T, mean;
  const float *W_ih, *b_ih, *W_hh, *b_hh;              /* GRUCell: [3M, 2M + D + T], [3M], [3M, M], [3M] */
  float *ws_aggr, *ws_h, *ws_gi, *ws_gh;               /* [R, 2M + D + T], [R, M], [R, 3M], [R, 3M] */
  float* out_mem; int64_t* out_lu;                     /* [R, M], [R] */
  int64_t* assoc; int64_t stamp;                       /* optional, as in tgmx_tgn_aggregate */
} tgmx_tgn_memory_fwd_t;
int tgmx_tgn_memory_forward(const tgmx_tgn_memory_fwd_t* args, tgmx_stream_t stream);

typedef struct tgmx_tconv_fwd {
  const float* x; int64_t U; int32_t in_ch;
  const int64_t* last_update_local;                    /* [U] */
  const int64_t* src; const int64_t* tgt; const int64_t* t; const float* msg; int64_t E; int32_t D, T;
  const float* tw; const float* tb;
  const float* W4; const float* b4; const float* W_edge; /* [4, H*C, in_ch], [4, H*C], [H*C, T + D] */
  int32_t H, C;
  float* edge_attr; float* qkvs; float* eproj;         /* [E, T + D], [4, U, H*C], [E, H*C] */
  int64_t* order; int64_t* seg_lo; int64_t* seg_hi;    /* [E], [U], [U] */
  void* sort_ws; size_t sort_ws_bytes; int32_t* status;
  /* optional (all three, or NULL: the segment-sort path): the counting grouping of the edges by target.  tgt_count [U] int32 must be
   * ZERO on entry and is left zero (the caller keeps it between calls: allocate once, zeroed); cursor [U], order_big [E] are scratch. */
  int32_t* tgt_count; int64_t* cursor; int64_t* order_big;
} tgmx_tconv_fwd_t;
int tgmx_tconv_forward(const tgmx_tconv_fwd_t* args, tgmx_stream_t stream);

/* ABI v7 -- the model side of one TGN batch as ONE call (examples/linkproppred/tgn.py:96-116 in inference order: memory(n_id) ->
 * GraphAttentionEmbedding -> memory.update_state): tgmx_tgn_memory_forward(mem), tgmx_tconv_forward(conv) with conv->x = mem->out_mem and
 * conv->last_update_local = mem->out_lu, then update_state with the rows `mem` just produced (TGNMemory.reuse_forward: tgmx_tgn_commit_assoc
 * over the batch's endpoints, mem->assoc / mem->stamp must be set) and tgmx_tgn_store_batch for the batch's n <= 1024 events.  The same
 * launches in the same order as the three module calls -- identical results -- issued back to back from C (the host side of a TGN batch
 * is ~10 us per launch when every launch is its own Python-mediated call). */
typedef struct tgmx_tgn_step {
  const tgmx_tgn_memory_fwd_t* mem;
  const tgmx_tconv_fwd_t* conv;                         /* NULL: no embedding (memory + update only) */
  const int32_t* src; const int32_t* dst; const int64_t* t; const float* raw; int32_t n;   /* the batch's events; raw [n, mem->D] */
  float* memory; int64_t* last_update; int32_t* reuse_status;                              /* written by the commit */
  int64_t log_base; int32_t* log_other; int64_t* log_t; float* log_raw;                    /* message log, rows [log_base, log_base + 2 n) */
  int64_t* st_lo_s; int32_t* st_cnt_s; int64_t* st_lo_d; int32_t* st_cnt_d;
} tgmx_tgn_step_t;
int tgmx_tgn_step(const tgmx_tgn_step_t* args, tgmx_stream_t stream);

/* The sampled edge list of one hop as the reference's TGN loop assembles it from torch ops
 * (examples/linkproppred/tgn.py:80-92): for every valid slot (nbr != -1), in slot order,
 *   edge_index[0][e] = local(seed of the row), edge_index[1][e] = local(nbr), edge_t[e] = nbr_t, edge_x[e] = nbr_x row,
 * local(v) = position of v in the sorted unique ids `uniq` (DeduplicationHook's global_to_local, tgm/hooks/dedup.py:60-66).
 * edge_index is [2, cap] (row 1 starts at cap), cap >= S * k; *count = number of edges (device); row_off: scratch [S + 1]. */
int tgmx_tgn_edge_list(const int32_t* seed, const int32_t* nbr, const int64_t* nbr_t, const float* nbr_x, int64_t S, int32_t k,
                       int32_t D, const int32_t* uniq, int64_t U, const int64_t* uniq_count /* device-side U, NULL = use U */,
                       int64_t cap, int64_t* row_off, int64_t* edge_index, int64_t* edge_t, float* edge_x, int64_t* count,
                       tgmx_stream_t stream);
/* The same with edge features BY ID (RecencyNeighborHook(edge_features='by_id'): the sampler publishes the store id of the edge behind every
 * slot, tgmx_recency_step_t.out_eid, instead of copying its feature row): edge_x[e] = edge_table[nbr_eid of the slot] -- the dense
 * [S, k, D] copies of recency.py:287-319 are never made (at the review shape: 11 MB per batch that the TGN model never reads beyond hop 0). */
int tgmx_tgn_edge_list_by_id(const int32_t* seed, const int32_t* nbr, const int64_t* nbr_t, const int32_t* nbr_eid, const float* edge_table,
                             int64_t S, int32_t k, int32_t D, const int32_t* uniq, int64_t U, const int64_t* uniq_count, int64_t cap,
                             int64_t* row_off, int64_t* edge_index, int64_t* edge_t, float* edge_x, int64_t* count, tgmx_stream_t stream);

/* torch.nn.GRUCell gates from gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh ([R, 3M], r|z|n). */
int tgmx_tgn_gru_gate(const float* gi, const float* gh, const float* h, int32_t M, int64_t R,
                      float* out, tgmx_stream_t stream);

/* memory[nodes[r]] = val[r]; last_update[nodes[r]] = lu[r] for rows with flag[r] (flag NULL: all). */
int tgmx_tgn_commit(const int32_t* nodes, const uint8_t* flag, const float* val, const int64_t* lu,
                    int32_t M, int32_t num_nodes, int64_t R, float* memory, int64_t* last_update,
                    tgmx_stream_t stream);

/* GraphAttentionEmbedding (tgn.py:14-40): edge_attr[e] = [Time2Vec(last_update_local[src[e]] - t[e]) | msg[e]] */
int tgmx_tconv_edge_attr(const int64_t* last_update_local, const int64_t* src, const int64_t* t,
                         const float* msg, const float* tw, const float* tb, int32_t T, int32_t D,
                         int64_t E, float* out, tgmx_stream_t stream);

/* Group n items by key[i] in [0, num_keys): order[] = the item ids stably sorted by key (what
 * argsort(stable=True) gives), seg_lo[k] / seg_hi[k] = the range of order[] holding key k.  Used for the incoming-edge
 * segments of TransformerConv (PyG scatter-softmax over edge_index[1]); replaces torch.sort + 2 x searchsorted.
 * Keys outside the range raise TGMX_ST_EDGE_RANGE in *status (and are clamped). */
size_t tgmx_segment_sort_workspace_bytes(int64_t n);
int tgmx_segment_sort(const int64_t* key, int64_t n, int32_t num_keys, int64_t* order, int64_t* seg_lo, int64_t* seg_hi,
                      void* workspace, size_t workspace_bytes, int32_t* status, tgmx_stream_t stream);

/* DGData.discretize's grouping, `_get_keep_indices` (tgm/data/dg_data.py:471-500), for one event group of n events:
 *   bucket[i]  = int32(floor(float64(time[i]) * factor))                                  (dg_data.py:469)
 *   key[i]     = bucket[i] * (max(id_key) + 1) + id_key[i],  id_key = id0 * (max(id0, id1) + 1) + id1 (or id0 when id1 is
 *                NULL) -- int32 arithmetic that wraps exactly like the reference's int32 tensors
 *   keep_pos   = positions of the first event of every distinct key in a STABLE sort of the keys, ascending
 *                (`reduce_op='first'`), *keep_count of them (device int64; keep_pos must hold n entries).
 * Ids must be >= 0.  workspace >= tgmx_discretize_workspace_bytes(n). */
size_t tgmx_discretize_workspace_bytes(int64_t n);
int tgmx_discretize_keep(const int64_t* time, const int32_t* id0, const int32_t* id1, int64_t n, double factor, int32_t* bucket,
                         int64_t* keep_pos, int64_t* keep_count, void* workspace, size_t workspace_bytes, tgmx_stream_t stream);

/* TransformerConv attention (third-party definition, PyG 2.6.1): for every target i,
 * out[i] += sum_j softmax_j(q_i.(k_j + e_ij)/sqrt(C)) (v_j + e_ij) per head; edges of target i are
 * order[seg_lo[i] .. seg_hi[i]) (edge ids sorted by target), src[e] = source j. */
int tgmx_tconv_attend(const float* q, const float* k, const float* v, const float* eproj,
                      const int64_t* order, const int64_t* src, const int64_t* seg_lo,
                      const int64_t* seg_hi, int64_t U, int32_t H, int32_t C, float scale, float* out,
                      const tgmx_dropout_t* drop /* training: dropout on the coefficients after the softmax, element
                                                    e * H + h (PyG TransformerConv dropout=; tgn.py:25-27 uses 0.1); NULL = off */,
                      tgmx_stream_t stream);

/* ------------------------------------------------------------------------
 * Discrete-time path (tgm/nn/encoder/tgcn.py): GCNConv normalisation + TGCN gates.
 * ------------------------------------------------------------------------ */

/* A[N, ld] (dense) = D^-1/2 (A + I) D^-1/2 with A[dst, src] = sum of the weights of edges src -> dst
 * (weight NULL = 1), self loops added where missing with weight `fill` (1, or 2 for improved=True),
 * degrees taken at the target: torch_geometric's gcn_norm (third-party to the reference).
 * workspace: 2 * N floats. */
int tgmx_gcn_norm_dense(const int64_t* src, const int64_t* dst, const float* weight, int64_t E,
                        int64_t N, float fill, int32_t add_self_loops, float* A, int64_t ld,
                        float* workspace, tgmx_stream_t stream);

/* out[N, 2C] = [a[:, :C] | b * sigmoid(gate_pre)]  (gate_pre NULL: no gating)   tgcn.py:118-149 */
int tgmx_tgcn_concat(const float* a, int64_t lda, const float* b, const float* gate_pre, int32_t C,
                     int64_t N, float* out, tgmx_stream_t stream);

/* out = sigmoid(u_pre) * H + (1 - sigmoid(u_pre)) * tanh(c_pre), n elements       tgcn.py:151-156 */
int tgmx_tgcn_output(const float* u_pre, const float* c_pre, const float* H, int64_t n, float* out,
                     tgmx_stream_t stream);

/* The whole TGCN cell forward (tgm/nn/encoder/tgcn.py:118-156, inference) as ONE call: tgmx_gcn_norm_dense, the two GEMMs of the three
 * convolutions over the shared A_hat (G = A_hat (X [W_u | W_r | W_c]^T) + [b_u | b_r | b_c]), per gate tgmx_tgcn_concat + its Linear, and
 * tgmx_tgcn_output -- the same launches in the same order as the entry points above called one by one (identical results), issued back to
 * back from C: a snapshot of a 255-node graph is 13 launches and the Python-composed sequence was host-bound at ~160 us.  Gate order
 * u, r, c everywhere.  Scratch: A [N, ldA] (ldA >= N, a multiple of 4), norm_ws [2 N], xwt [3 C, N], G [N, 3 C], cat [N, 2 C], pre [3][N, C]. */
typedef struct tgmx_tgcn_fwd {
  const float* x; int64_t N; int32_t in_ch, C;          /* node features [N, in_ch]; C = out_channels */
  const int64_t* src; const int64_t* dst; const float* edge_w; int64_t E; float fill; int32_t add_self_loops;
  const float* W3; const float* b3;                     /* stacked GCN weights [3 C, in_ch] and biases [3 C] */
  const float* lin_w[3]; const float* lin_b[3];         /* linear_{u,r,c}: [C, 2 C], [C] */
  const float* H;                                       /* [N, C] recurrent state */
  float* A; int64_t ldA; float* norm_ws; float* xwt; float* G; float* cat; float* pre[3];
  float* out;                                           /* [N, C] */
  int32_t idx32;                                        /* != 0: src / dst point at int32 ids (a DGBatch's edge_src / edge_dst as they are) */
} tgmx_tgcn_fwd_t;
int tgmx_tgcn_forward(const tgmx_tgcn_fwd_t* args, tgmx_stream_t stream);

/* Backward of the TGCN cell (tgcn.py:151-156 under loss.backward(), examples/nodeproppred/tgcn.py:92).  out = U H + (1 - U) tanh(c_pre),
 * U = sigmoid(u_pre):  du_pre = dout (H - Cc) U (1 - U), dc_pre = dout (1 - U) (1 - Cc^2), dH = dout U (n = N * C elements each). */
int tgmx_tgcn_gate_backward(const float* dout, const float* u_pre, const float* c_pre, const float* H, int64_t n, float* du_pre,
                            float* dc_pre, float* dH, tgmx_stream_t stream);
/* The gates' contributions to dH and the reset gate's pre-activation gradient.  dcat_x = dx_pre W_x are [N, 2C] (the gradient of the
 * gate's input [conv_x(X) | H] resp. [conv_c(X) | H R]); any of the three may be NULL (skipped):
 *   dcat_c: dr_pre = dcat_c[:, C:] H R (1 - R) (written when dr_pre != NULL), dH += dcat_c[:, C:] R;   dcat_u, dcat_r: dH += dcat[:, C:]. */
int tgmx_tgcn_reset_backward(const float* dcat_c, const float* dcat_u, const float* dcat_r, const float* r_pre, const float* H, int32_t C,
                             int64_t N, float* dr_pre, float* dH, tgmx_stream_t stream);

/* RandomNegativeEdgeSamplerHook (tgm/hooks/negatives/sampler.py:45-65): out_neg[i] uniform in [low, high), out_time =
 * copy of time_in (the reference: torch.randint + edge_time.clone()), one launch.  Counter-based generator keyed by
 * (seed, call, i): same distribution as the reference, a different stream than torch's. */
int tgmx_random_negatives(int32_t low, int32_t high, int64_t n, uint64_t seed, uint64_t call, int32_t* out_neg,
                          const int64_t* time_in, int64_t n_time, int64_t* out_time, tgmx_stream_t stream);
/* ABI v4: the same draws starting at index `index0` of the call's sequence (a rank's share of a sharded batch: index0 = the
 * share's offset in the batch); tgmx_random_negatives is index0 = 0. */
int tgmx_random_negatives_at(int32_t low, int32_t high, int64_t n, uint64_t seed, uint64_t call, int64_t index0, int32_t* out_neg,
                          const int64_t* time_in, int64_t n_time, int64_t* out_time, tgmx_stream_t stream);

/* DeduplicationHook (tgm/hooks/dedup.py:35-67): the sorted unique ids of up to 16 int32 id arrays (edge endpoints, extra
 * seed attributes, every hop's neighbor ids); -1 (padded slot) is skipped inside the kernel, ids outside [0, num_nodes)
 * raise TGMX_ST_SEED_RANGE.  out_ids needs room for min(total ids, num_nodes); *out_count (device) receives the number
 * written.  `parts` / `part_sizes` are HOST arrays of device pointers / lengths.  workspace: 256-byte aligned,
 * tgmx_unique_ids_workspace_bytes(num_nodes) bytes, ALL ZERO when first handed to the library (the node bitmap: every call
 * leaves it zero again, so there is no memset per call). */
size_t tgmx_unique_ids_workspace_bytes(int32_t num_nodes);
int tgmx_unique_ids(const int32_t* const* parts, const int64_t* part_sizes, int32_t num_parts, int32_t num_nodes,
                    void* workspace, int32_t* out_ids, int64_t* out_count, int32_t* status, tgmx_stream_t stream);

/* Group n <= 1024 int32 ids in one launch: stable sort by id, the permutation (sorted position -> input index), every
 * position's run [run_lo, run_hi) and a first-of-run flag; any output may be NULL.  Replaces the torch.sort +
 * searchsorted glue around the TGN message store and commit (tgm/nn/encoder/tgn.py:165-177, 218-229). */
int tgmx_group_ids(const int32_t* ids, int32_t n, int32_t* sorted, int64_t* perm, int64_t* run_lo, int64_t* run_hi,
                   uint8_t* first, tgmx_stream_t stream);
/* The same for any n (a 4096-edge batch groups 8192 endpoint ids): one stable rocPRIM radix sort (signed int32 order, like
 * torch.sort(stable=True)) + one finishing launch.  workspace: 256-byte aligned, tgmx_group_ids_workspace_bytes(n) bytes. */
size_t tgmx_group_ids_workspace_bytes(int64_t n);
int tgmx_group_ids_large(const int32_t* ids, int64_t n, int32_t* sorted, int64_t* perm, int64_t* run_lo, int64_t* run_hi,
                         uint8_t* first, void* workspace, size_t workspace_bytes, tgmx_stream_t stream);

/* ---- TGN backward building blocks (training; composed by tgm_amd/nn/_tgn_train.py).  The reference trains through
 * torch autograd (examples/linkproppred/tgn.py:97-118); memory / last_update are buffers, so the parameters reached are
 * the shared Time2Vec (tgm/nn/modules/time_encoding.py), the GRU cell and the TransformerConv projections.  The dense
 * projections differentiate through tgmx_sgemm_nt / tgmx_sgemm_tn / tgmx_colsum. ---- */

/* d(gi), d(gh) [R, 3M] of tgmx_tgn_gru_gate (torch.nn.GRUCell gate arithmetic); h is a buffer row (no dh) */
int tgmx_tgn_gru_gate_backward(const float* gi, const float* gh, const float* h, const float* dout, int32_t M, int64_t R,
                               float* dgi, float* dgh, tgmx_stream_t stream);
/* backward of tgmx_tgn_aggregate w.r.t. the Time2Vec parameters: part[r, :T] = d(tw), part[r, T:2T] = d(tb) contribution
 * of row r (same event selection as the forward; sum the rows with tgmx_colsum).  d_aggr: [R, 2M + D + T].  The row_*
 * arrays are per-row snapshots of the node's message windows and last_update taken at forward time: the reference's
 * training loop calls update_state before loss.backward() (examples/linkproppred/tgn.py:111-116). */
int tgmx_tgn_aggregate_backward(int64_t R, const int64_t* row_lo_s, const int32_t* row_cnt_s, const int64_t* row_lo_d,
                                const int32_t* row_cnt_d, const int64_t* row_last_update, const int64_t* log_t, int32_t M,
                                int32_t D, const float* tw, const float* tb, int32_t T, int32_t mean, const float* d_aggr,
                                float* part, tgmx_stream_t stream);
/* backward of tgmx_tconv_edge_attr w.r.t. the Time2Vec parameters: part [E, 2T] (d_attr: [E, T + D]) */
int tgmx_tconv_edge_attr_backward(const int64_t* last_update_local, const int64_t* src, const int64_t* t, const float* tw,
                                  const float* tb, const float* d_attr, int32_t T, int32_t D, int64_t E, float* part,
                                  tgmx_stream_t stream);
/* backward of tgmx_tconv_attend: dq written, dk / dv zero-initialised by the caller and accumulated (float atomics),
 * de [E, H*C] written for every edge.  C <= 64. */
int tgmx_tconv_attend_backward(const float* q, const float* k, const float* v, const float* eproj, const int64_t* order,
                               const int64_t* src, const int64_t* seg_lo, const int64_t* seg_hi, int64_t U, int32_t H,
                               int32_t C, float scale, const float* dout, float* dq, float* dk, float* dv, float* de,
                               const tgmx_dropout_t* drop /* the forward's, NULL = off */, tgmx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TGM_AMD_H */
