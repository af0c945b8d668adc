// Wave launch rate as a function of the VGPR allocation (4 480 one-wave workgroups per us whatever the allocation):
// hipcc --offload-arch=gfx950 -O3 -w -o launch_rate tools/launch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define K(NV)                                                                  \
  __global__ void k##NV(float* out) {                                           \
    asm volatile("v_mov_b32 v" #NV ", 0" ::: "v" #NV);                          \
    if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1.f;             \
  }
K(15) K(31) K(63) K(95) K(127) K(167) K(215) K(255)
int main() {
  float* out; hipMalloc(&out, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, double waves) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us  -> %7.0f waves/us\n", name, ms * 1e3, waves / (ms * 1e3));
  };
  const int N = 1 << 18;
#define R(NV) \
  run("v" #NV " 64-thread WGs", [&] { hipLaunchKernelGGL(k##NV, dim3(N), dim3(64), 0, 0, out); }, N); \
  run("v" #NV " 256-thread WGs", [&] { hipLaunchKernelGGL(k##NV, dim3(N / 4), dim3(256), 0, 0, out); }, N);
  R(15) R(31) R(63) R(95) R(127) R(167) R(215) R(255)
  return 0;
}
