// How fast does the GPU start workgroups?  Empty kernels over grid sizes and workgroup sizes.
//   hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip -o dispatch_rate && ./dispatch_rate
#include <cstdio>
#include <hip/hip_runtime.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void nothing(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
template <int LDS>
__global__ void nothing_lds(int* p) { __shared__ int s[LDS / 4]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (p && s[(threadIdx.x + 1) % 64] == 99999) *p = 1; }
// a little work per thread: ~n dependent FMAs
__global__ void some_work(float* p, int n) { float a = threadIdx.x * 1e-3f; for (int i = 0; i < n; i++) a = fmaf(a, 1.0001f, 0.5f); if (a == 12345.f) *p = a; }

template <class F>
float time_us(F f, int reps = 50) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; i++) f();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; i++) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  int* d; CK(hipMalloc(&d, 64));
  for (int block : {64, 256, 1024}) {
    for (int wgs : {256, 1024, 4096, 8192, 16384, 65536}) {
      const float t = time_us([&] { hipLaunchKernelGGL(nothing, dim3(wgs), dim3(block), 0, 0, d); });
      printf("empty      block %4d  workgroups %6d  %8.2f us  %6.2f ns/workgroup\n", block, wgs, t, 1e3f * t / wgs);
    }
  }
  for (int wgs : {1024, 8192}) {
    float t = time_us([&] { hipLaunchKernelGGL(nothing_lds<12800>, dim3(wgs), dim3(256), 0, 0, d); });
    printf("12.5KB LDS block  256  workgroups %6d  %8.2f us  %6.2f ns/workgroup\n", wgs, t, 1e3f * t / wgs);
    t = time_us([&] { hipLaunchKernelGGL(nothing_lds<65536>, dim3(wgs), dim3(1024), 0, 0, d); });
    printf("64KB LDS   block 1024  workgroups %6d  %8.2f us  %6.2f ns/workgroup\n", wgs, t, 1e3f * t / wgs);
  }
  for (int n : {1000, 10000}) {
    for (int wgs : {1024, 4096, 8192}) {
      const float t = time_us([&] { hipLaunchKernelGGL(some_work, dim3(wgs), dim3(256), 0, 0, (float*)d, n); });
      printf("%5d FMAs block  256  workgroups %6d  %8.2f us\n", n, wgs, t);
    }
  }
  fflush(stdout);
  return 0;
}
