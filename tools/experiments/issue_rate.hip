// How fast does one SIMD of gfx950 issue vector instructions?  Dependent chains and independent ones, for 1..8
// waves per SIMD, for a few instruction kinds of the blend kernels (fma, exp2, rcp, DPP add, select, LDS read).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 issue_rate.hip -o issue_rate && ./issue_rate
// Prints shader-clock cycles (s_memtime, read by the kernel itself) per wave-instruction per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 4096;

enum Kind { FMA = 0, EXP, RCP, DPP, SEL, LDS, CMPSEL };

template <int KIND, int ILP>
__global__ void __launch_bounds__(1024) chain(float* out, float seed, long long* cycles) {
  __shared__ float s[1024];
  s[threadIdx.x] = seed;
  __syncthreads();
  const long long t0 = clock64();
  float v[ILP];
#pragma unroll
  for (int k = 0; k < ILP; k++) v[k] = seed + (float)(threadIdx.x + k);
  const float a = 1.0000001f, b = 1e-9f;
  int idx = threadIdx.x;
#pragma unroll 16
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) {
      if (KIND == FMA) v[k] = __builtin_fmaf(v[k], a, b);
      if (KIND == EXP) v[k] = __builtin_amdgcn_exp2f(v[k] * 1e-3f);
      if (KIND == RCP) v[k] = __builtin_amdgcn_rcpf(v[k] + 2.0f);
      if (KIND == DPP) v[k] = v[k] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0xB1, 0xf, 0xf, false));
      if (KIND == SEL) v[k] = v[k] > 0.5f ? v[k] * a : b;
      if (KIND == CMPSEL) { const bool c = v[k] > 0.5f; v[k] = c ? a : v[k]; }
      if (KIND == LDS) { idx = __float_as_int(s[idx & 1023]) & 1023; v[k] += (float)idx; }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < ILP; k++) r += v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + (float)idx;
  __syncthreads();
  if (threadIdx.x == 0) cycles[blockIdx.x] = clock64() - t0;
}

template <int KIND, int ILP>
double run(int waves_per_simd, float* out, long long* cycles, int cus) {
  // waves_per_simd * 4 waves per CU: one block of up to 1024 threads per CU, two of them for 8 waves per SIMD
  const int per_cu = waves_per_simd > 4 ? 2 : 1;
  const int threads = waves_per_simd * 4 * 64 / per_cu, blocks = cus * per_cu;
  hipLaunchKernelGGL((chain<KIND, ILP>), dim3(blocks), dim3(threads), 0, 0, out, 0.75f, cycles);
  hipLaunchKernelGGL((chain<KIND, ILP>), dim3(blocks), dim3(threads), 0, 0, out, 0.75f, cycles);
  if (hipDeviceSynchronize() != hipSuccess) return -1.0;
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cycles, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (long long c : h) sum += (double)c;
  return sum / blocks;     // mean cycles a block took
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  float* out;
  long long* cycles;
  CK(hipMalloc(&out, (size_t)cus * 1024 * 4 * 2));
  CK(hipMalloc(&cycles, (size_t)cus * 2 * sizeof(long long)));
  printf("%s: %d CUs, %.2f GHz nominal\n", p.gcnArchName, cus, ghz);
  printf("cycles per wave-instruction per SIMD (one block per CU; the instruction itself plus what the loop adds)\n");
  printf("%-28s %8s %8s %8s %8s\n", "kind, chains per lane", "1 wave", "2 waves", "4 waves", "8 waves");
#define ROW(KIND, ILP, name, per)                                                                   \
  {                                                                                                 \
    printf("%-28s", name);                                                                          \
    for (int w : {1, 2, 4, 8}) {                                                                    \
      const double cyc = run<KIND, ILP>(w, out, cycles, cus);                                       \
      const double instr = (double)kIters * ILP * per * w; /* per SIMD */                           \
      printf(" %8.2f", cyc / instr);                                                                \
    }                                                                                               \
    printf("\n");                                                                                   \
  }
  ROW(FMA, 1, "fma, 1 dependent chain", 1)
  ROW(FMA, 2, "fma, 2 chains", 1)
  ROW(FMA, 4, "fma, 4 chains", 1)
  ROW(FMA, 8, "fma, 8 chains", 1)
  ROW(EXP, 1, "mul + exp2, 1 chain", 2)
  ROW(EXP, 4, "mul + exp2, 4 chains", 2)
  ROW(RCP, 1, "add + rcp, 1 chain", 2)
  ROW(RCP, 4, "add + rcp, 4 chains", 2)
  ROW(DPP, 1, "add with DPP, 1 chain", 1)
  ROW(DPP, 4, "add with DPP, 4 chains", 1)
  ROW(SEL, 1, "cmp + mul + select, 1 chain", 3)
  ROW(SEL, 4, "cmp + mul + select, 4 ch.", 3)
  ROW(LDS, 1, "ds_read chase (+3 valu)", 1)
  fflush(stdout);
  return 0;
}
