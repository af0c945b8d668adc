// Do two HIP streams of one process overlap chains of small dependent kernels on this stack?  Each kernel is ONE workgroup busy for ~T us
// (a clock spin); a chain is N of them back to back on a stream.  One stream alone takes N * (T + gap); two streams overlap perfectly if the
// pair takes the same, not at all if it takes twice that.  Also: a chain of small kernels beside a chip-filling kernel on the other stream.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/stream_overlap tools/stream_overlap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 0 && cycles < 0) *sink = 1;
}
__global__ void fill(float* p, long long n, int iters) {  // every CU busy: a grid-stride fma loop
  float a = 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    for (int k = 0; k < iters; ++k) a = a * 1.0001f + p[i];
  if (a == 12345.f) p[0] = a;
}
int main() {
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  float* buf; hipMalloc(&buf, 64 << 20);
  hipMemset(buf, 0, 64 << 20);
  const long long T = 500;  // wall_clock64 ticks at 100 MHz: 500 = 5 us
  const int N = 200;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto chain = [&](hipStream_t s, int wgs) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, s, T, nullptr); };
  for (int wgs : {1, 64, 2048}) {
    chain(s1, wgs); hipDeviceSynchronize();
    double t0 = now(); chain(s1, wgs); hipDeviceSynchronize(); double one = now() - t0;
    t0 = now(); for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, s1, T, nullptr); hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, s2, T, nullptr); }
    hipDeviceSynchronize(); double two = now() - t0;
    printf("{\"kernel\": \"5 us spin, %d workgroups of one wave\", \"chain\": %d, \"one_stream_us_per_kernel\": %.2f, \"two_streams_us_per_pair\": %.2f, \"overlap\": %.2f}\n", wgs, N,
           one / N, two / N, 2.0 - two / one);
  }
  // a chain of small kernels beside a chip-filling kernel
  const long long n = 16 << 20;
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, s2, buf, n, 40); hipDeviceSynchronize();
  double t0 = now(); hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, s2, buf, n, 40); hipDeviceSynchronize(); double big = now() - t0;
  t0 = now(); for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, T, nullptr); hipDeviceSynchronize(); double small = now() - t0;
  t0 = now(); hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, s2, buf, n, 40);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, T, nullptr);
  hipDeviceSynchronize(); double both = now() - t0;
  printf("{\"big_kernel_us\": %.1f, \"chain_of_20_small_us\": %.1f, \"both_on_two_streams_us\": %.1f}\n", big, small, both);
  return 0;
}
