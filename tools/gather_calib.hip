// What rocprofv3's FETCH_SIZE / WRITE_SIZE read for KNOWN byte counts in the sampler's own access patterns (missing 5: the
// guide's x2 FETCH_SIZE correction is calibrated for 16 B/lane streaming reads only).  Every kernel moves exactly 1 GiB in and 1 GiB out:
// 16-byte units, unit u of chunk c = u / UNITS is read from table[perm(c) * CHUNK + (u % UNITS) * 16] and written to out[u * 16]
// (perm = multiplication by an odd constant modulo a power of two: every chunk is read once, nothing is re-read, the 4 GiB table is
// 16 x the Infinity Cache).  CHUNK = 16: one ring record; 64: one D = 16 feature row; 320: a node's window of B = 20 records;
// 1280: a node's 20 feature rows (ring_x[n * B ...]); stream: CHUNK = the whole buffer (the guide's calibration case).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gather_calib tools/gather_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -- tools/bin/gather_calib ; rocprofv3 --pmc WRITE_SIZE -- tools/bin/gather_calib   (tools/gpu_calib_r5.sh)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct alignas(16) U16 { uint32_t a, b, c, d; };
template <int CHUNK>
__global__ __launch_bounds__(256) void gather_calib(const U16* __restrict__ table, U16* __restrict__ out, uint64_t n_units, uint64_t chunk_mask) {
  constexpr uint64_t UNITS = CHUNK / 16;
  const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_units) return;
  const uint64_t c = u / UNITS, w = u % UNITS;
  const uint64_t pc = (c * 0x9E3779B97F4A7C15ull + 12345) & chunk_mask;  // odd multiplier: a permutation of [0, mask]
  out[u] = table[pc * UNITS + w];
}
__global__ __launch_bounds__(256) void stream_calib(const U16* __restrict__ table, U16* __restrict__ out, uint64_t n_units) {
  const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n_units) out[u] = table[u];
}
__global__ __launch_bounds__(256) void read_only_calib(const U16* __restrict__ table, U16* __restrict__ out, uint64_t n_units) {
  const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_units) return;
  const U16 v = table[u];
  if (v.a == 0xdeadbeefu && v.b == 17u) out[u] = v;  // never true: the read stays, nothing is written
}
int main() {
  const uint64_t table_bytes = 4ull << 30, n_units = (1ull << 30) / 16;
  U16 *table, *out;
  if (hipMalloc(&table, table_bytes) != hipSuccess || hipMalloc(&out, n_units * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(table, 1, table_bytes); hipMemset(out, 0, n_units * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid((unsigned)(n_units / 256)), block(256);
  auto run = [&](const char* name, int chunk, auto launch) {
    for (int rep = 0; rep < 3; ++rep) {  // three launches each: the counters are averaged per kernel name
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("{\"kernel\": \"%s\", \"chunk_bytes\": %d, \"read_bytes\": %llu, \"write_bytes\": %llu, \"us\": %.1f, \"GBps_read_plus_write\": %.0f}\n", name, chunk,
             (unsigned long long)(n_units * 16), (unsigned long long)(chunk == -1 ? 0 : n_units * 16), ms * 1e3, (chunk == -1 ? 1 : 2) * n_units * 16 / (ms * 1e-3) / 1e9);
    }
  };
  auto mask = [&](int chunk) { uint64_t n = 1; while (n * 2 * chunk <= table_bytes) n *= 2; return n - 1; };
  run("stream_calib", 0, [&] { hipLaunchKernelGGL(stream_calib, grid, block, 0, 0, table, out, n_units); });
  run("read_only_calib", -1, [&] { hipLaunchKernelGGL(read_only_calib, grid, block, 0, 0, table, out, n_units); });
  run("gather_calib<16>", 16, [&] { hipLaunchKernelGGL(gather_calib<16>, grid, block, 0, 0, table, out, n_units, mask(16)); });
  run("gather_calib<64>", 64, [&] { hipLaunchKernelGGL(gather_calib<64>, grid, block, 0, 0, table, out, n_units, mask(64)); });
  run("gather_calib<320>", 320, [&] { hipLaunchKernelGGL(gather_calib<320>, grid, block, 0, 0, table, out, n_units, mask(320)); });
  run("gather_calib<1280>", 1280, [&] { hipLaunchKernelGGL(gather_calib<1280>, grid, block, 0, 0, table, out, n_units, mask(1280)); });
  return 0;
}
