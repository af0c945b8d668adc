// DeduplicationHook (tgm/hooks/dedup.py:35-67): sorted unique node ids of a batch.  The reference concatenates the id
// tensors (after boolean-masking the pad slots of every hop) and calls torch.unique (a sort).  Node ids are bounded by
// num_nodes, so a bitmap does it in O(ids + num_nodes / 32) with the result already sorted: mark (atomicOr, pads
// skipped in the kernel -- no boolean indexing, no device -> host sync per hop), per-word popcount scanned in one
// workgroup, then every word scatters its set bits to their final positions.
#include "common.h"

namespace tgmx {

constexpr int kMaxParts = 16;

struct MarkArgs {
  const int32_t* part[kMaxParts];
  long long end[kMaxParts];  // exclusive end offset of part p in the concatenation
  unsigned int* bitmap;
  int32_t* status;
  int parts, N;
  EdgeListScan rider;  // nbr != NULL: the LAST workgroup of the launch scans the edge list's rows instead of marking (common.h)
};

__global__ __launch_bounds__(256) void dedup_mark_kernel(const MarkArgs a) {
  if (a.rider.nbr && blockIdx.x == gridDim.x - 1) {
    edge_list_scan_body(a.rider.nbr, a.rider.S, a.rider.k, a.rider.row_off, a.rider.count);
    return;
  }
  const long long total = a.end[a.parts - 1];
  const long long step = (long long)(gridDim.x - (a.rider.nbr ? 1 : 0)) * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
    const int32_t* p = a.part[0];
    long long base = 0;
#pragma unroll
    for (int q = 1; q < kMaxParts; ++q) {
      if (q < a.parts && i >= a.end[q - 1]) {
        p = a.part[q];
        base = a.end[q - 1];
      }
    }
    const int id = p[i - base];
    if (id == -1) continue;  // padded neighbor slot
    if (id < 0 || id >= a.N) {
      atomicOr(a.status, TGMX_ST_SEED_RANGE);
      continue;
    }
    atomicOr(&a.bitmap[id >> 5], 1u << (id & 31));
  }
}

// One workgroup walks the bitmap (num_nodes / 32 words: ~11 k at review scale) 1024 words at a time: exclusive prefix sum
// of the popcounts, every set bit written to its slot of the ascending output -- and the word cleared again, so the bitmap
// is all-zero when the call ends (no memset per call; it only has to be zero when the workspace is first used).
__global__ __launch_bounds__(1024) void dedup_scan_compact_kernel(unsigned int* __restrict__ bitmap, long long words, int32_t* __restrict__ out,
                                                                  int64_t* __restrict__ count) {
  // One workgroup walks the bitmap in passes of 16 K words (512 K node ids): a thread owns kPer CONSECUTIVE words of a pass, loaded
  // up front as independent 16-byte reads -- one pass covers the review-shaped graph (350 k nodes: 11 k words).  (Round 3 read one
  // word per thread per trip: eleven dependent trips of load -> scan -> three barriers, 14 us of latency for 44 KB.)
  constexpr int kPer = 16;
  __shared__ int wave_tot[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  const bool vec = (reinterpret_cast<uintptr_t>(bitmap) & 15) == 0;
  for (long long w0 = 0; w0 < words; w0 += 1024 * kPer) {
    const long long wb = w0 + (long long)tid * kPer;
    unsigned int bits[kPer];
    if (vec && wb + kPer <= words) {
#pragma unroll
      for (int q = 0; q < kPer / 4; ++q) {
        const uint4 v4 = reinterpret_cast<const uint4*>(bitmap + wb)[q];
        bits[4 * q] = v4.x; bits[4 * q + 1] = v4.y; bits[4 * q + 2] = v4.z; bits[4 * q + 3] = v4.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < kPer; ++q) bits[q] = wb + q < words ? bitmap[wb + q] : 0u;
    }
    int v = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) v += __popc(bits[q]);
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = carry_s;
    for (int q = 0; q < wave; ++q) before += wave_tot[q];
    if (v) {
      int pos = before + incl - v;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        unsigned int b32 = bits[q];
        if (b32) {
          bitmap[wb + q] = 0u;
          while (b32) {
            const int b = __ffs(b32) - 1;
            out[pos++] = (int)((wb + q) * 32 + b);
            b32 &= b32 - 1;
          }
        }
      }
    }
    __syncthreads();
    if (tid == 1023) carry_s = before + incl;
    __syncthreads();
  }
  if (tid == 0) *count = carry_s;
}

// RandomNegativeEdgeSamplerHook (tgm/hooks/negatives/sampler.py:45-65): neg[i] uniform in [low, high), neg_time = copy of
// the batch's edge times -- the reference's randint + clone as one launch, counter-based generator (seed, call, i)
__global__ __launch_bounds__(256) void random_negatives_kernel(int32_t* __restrict__ neg, long long n, int low, unsigned range,
                                                               unsigned long long seed, unsigned long long call, unsigned long long i0,
                                                               const int64_t* __restrict__ t_in, int64_t* __restrict__ t_out,
                                                               long long nt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) neg[i] = negative_draw(seed, call, i0 + (unsigned long long)i, low, range);
  if (i < nt) t_out[i] = t_in[i];
}

}  // namespace tgmx

using namespace tgmx;

extern "C" size_t tgmx_unique_ids_workspace_bytes(int32_t num_nodes) {
  const size_t words = ((size_t)(num_nodes > 0 ? num_nodes : 0) + 31) / 32;
  return words * 8 + 512;  // bitmap + per-word prefix
}

extern "C" int tgmx_unique_ids(const int32_t* const* parts, const int64_t* part_sizes, int32_t num_parts, int32_t num_nodes,
                               void* workspace, int32_t* out_ids, int64_t* out_count, int32_t* status, tgmx_stream_t stream) {
  return tgmx_internal_unique_ids(parts, part_sizes, num_parts, num_nodes, workspace, out_ids, out_count, status, nullptr, stream);
}

int tgmx_internal_unique_ids(const int32_t* const* parts, const int64_t* part_sizes, int32_t num_parts, int32_t num_nodes, void* workspace,
                             int32_t* out_ids, int64_t* out_count, int32_t* status, const EdgeListScan* rider, tgmx_stream_t stream) {
  TGMX_REQUIRE(num_parts > 0 && num_parts <= kMaxParts && num_nodes > 0, "unique_ids: %d parts (1..%d), num_nodes=%d", num_parts, kMaxParts,
               num_nodes);
  TGMX_REQUIRE(parts && part_sizes && workspace && out_ids && out_count && status, "unique_ids: null pointer");
  TGMX_REQUIRE(((uintptr_t)workspace & 255) == 0, "unique_ids: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const long long words = ((long long)num_nodes + 31) / 32;
  unsigned int* bitmap = reinterpret_cast<unsigned int*>(workspace);
  MarkArgs m{};
  long long total = 0;
  for (int p = 0; p < num_parts; ++p) {
    TGMX_REQUIRE(part_sizes[p] >= 0 && (part_sizes[p] == 0 || parts[p]), "unique_ids: part %d", p);
    total += part_sizes[p];
    m.part[p] = parts[p];
    m.end[p] = total;
  }
  m.bitmap = bitmap; m.status = status; m.parts = num_parts; m.N = num_nodes;
  const bool ride = rider && rider->nbr && rider->S > 0;
  if (ride) m.rider = *rider;
  if (total > 0 || ride) {
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dedup_mark_kernel, dim3((unsigned)blocks + (ride ? 1u : 0u)), dim3(256), 0, st, m);
  }
  hipLaunchKernelGGL(dedup_scan_compact_kernel, dim3(1), dim3(1024), 0, st, bitmap, words, out_ids, out_count);
  TGMX_CHECK_LAUNCH("unique_ids");
  return TGMX_OK;
}

extern "C" int tgmx_random_negatives_at(int32_t low, int32_t high, int64_t n, uint64_t seed, uint64_t call, int64_t index0, int32_t* out_neg,
                                        const int64_t* time_in, int64_t n_time, int64_t* out_time, tgmx_stream_t stream) {
  TGMX_REQUIRE(low < high && n >= 0 && n_time >= 0 && index0 >= 0, "random_negatives: bad arguments low=%d high=%d n=%lld", low, high, (long long)n);
  const long long work = n > n_time ? n : n_time;
  if (work == 0) return TGMX_OK;
  TGMX_REQUIRE((n == 0 || out_neg) && (n_time == 0 || (time_in && out_time)), "random_negatives: null pointer");
  hipLaunchKernelGGL(random_negatives_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out_neg, (long long)n,
                     low, (unsigned)((long long)high - low), (unsigned long long)seed, (unsigned long long)call, (unsigned long long)index0, time_in,
                     out_time, (long long)n_time);
  TGMX_CHECK_LAUNCH("random_negatives");
  return TGMX_OK;
}

extern "C" int tgmx_random_negatives(int32_t low, int32_t high, int64_t n, uint64_t seed, uint64_t call, int32_t* out_neg,
                                     const int64_t* time_in, int64_t n_time, int64_t* out_time, tgmx_stream_t stream) {
  return tgmx_random_negatives_at(low, high, n, seed, call, 0, out_neg, time_in, n_time, out_time, stream);
}
