// The issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 (independent accumulators, 1 or 2 waves per SIMD) and what SQ_VALU_MFMA_BUSY_CYCLES reads for
// a saturated loop: hipcc --offload-arch=gfx950 -O3 -w -o mfma_rate tools/mfma_rate.hip; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using floatx4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;
using floatx16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
  floatx4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k32(float* out, int iters, float a, float b) {
  floatx16 acc[2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][3];
}
int main() {
  float* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double mfmas_per_wave, int waves_per_simd) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.1f us  -> %6.2f ns per MFMA per SIMD (%d waves/SIMD)\n", name, ms * 1e3, ms * 1e6 / (mfmas_per_wave * waves_per_simd), waves_per_simd);
  };
  const int it = 20000;
  run("16x16x4 f32, 4 acc, 1 wave/SIMD", [&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(256), 0, 0, out, it, 1.f, 2.f); }, it * 16.0, 1);
  run("16x16x4 f32, 4 acc, 2 waves/SIMD", [&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(512), 0, 0, out, it, 1.f, 2.f); }, it * 16.0, 2);
  run("16x16x4 f32, 1 acc (dependent)", [&] { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(256), 0, 0, out, it, 1.f, 2.f); }, it * 4.0, 1);
  run("16x16x4 f32, 3 acc, 2 waves/SIMD", [&] { hipLaunchKernelGGL(k16<3>, dim3(256), dim3(512), 0, 0, out, it, 1.f, 2.f); }, it * 12.0, 2);
  run("16x16x4 f32, 4 acc, 197 CUs", [&] { hipLaunchKernelGGL(k16<4>, dim3(197), dim3(512), 0, 0, out, it, 1.f, 2.f); }, it * 16.0, 2);
  run("32x32x2 f32, 2 acc, 1 wave/SIMD", [&] { hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, out, it, 1.f, 2.f); }, it * 8.0, 1);
  return 0;
}
