// What do LDS instructions cost a CU of gfx950, and does it depend on how many lanes take part?  32 waves per CU
// (8 per SIMD) each issue the same instruction over and over; the table gives cycles of the CU per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 lds_rate.hip -o lds_rate && ./lds_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 2048, kPer = 8;

enum Mode { W32_ALL = 0, W32_LOW_HALF, W32_EVEN, W32_ONE, R128_ALL, R128_LOW_HALF, R64_BCAST_ALL, R64_BCAST_ONE, R2X64_BCAST_ALL,
            R32_ALL, ADDTID_ALL, W2X32_ALL, W64_ALL, W128_ALL, R2X32_ALL };

template <int MODE>
__global__ void __launch_bounds__(1024) lds_loop(float* out, long long* cycles) {
  __shared__ __align__(16) float s[16 * 1024 / 4 * 4];   // 64 KB: 4 KB per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) s[i] = (float)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)(s + wave * 1024));
  const unsigned a32 = base + 4u * lane;                  // one bank per lane
  const unsigned a128 = base + 16u * lane;                // 16 bytes per lane, consecutive
  float acc = 0.f;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f4 v4 = {0.f, 0.f, 0.f, 0.f};
  f2 v2 = {0.f, 0.f}, w2 = v2;
  const float val = (float)lane;
  bool on = true;
  if (MODE == W32_LOW_HALF || MODE == R128_LOW_HALF) on = lane < 32;
  if (MODE == W32_EVEN) on = (lane & 1) == 0;
  if (MODE == W32_ONE || MODE == R64_BCAST_ONE) on = lane == 0;
  const long long t0 = clock64();
  if (on) {
    for (int it = 0; it < kIters; it++) {
#pragma unroll
      for (int k = 0; k < kPer; k++) {
        if (MODE <= W32_ONE) asm volatile("ds_write_b32 %0, %1 offset:0" ::"v"(a32), "v"(val) : "memory");
        if (MODE == ADDTID_ALL) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:0" ::"v"(val), "s"(base) : "m0", "memory");
        if (MODE == R128_ALL || MODE == R128_LOW_HALF) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(a128) : "memory");
        if (MODE == R64_BCAST_ALL || MODE == R64_BCAST_ONE) asm volatile("ds_read_b64 %0, %1" : "=v"(v2) : "v"(base) : "memory");
        if (MODE == R2X64_BCAST_ALL) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v4) : "v"(base) : "memory");
        if (MODE == W2X32_ALL) asm volatile("ds_write2_b32 %0, %1, %1 offset0:0 offset1:64" ::"v"(a32), "v"(val) : "memory");
        if (MODE == W64_ALL) asm volatile("ds_write_b64 %0, %1" ::"v"(base + 8u * lane), "v"(v2) : "memory");
        if (MODE == W128_ALL) asm volatile("ds_write_b128 %0, %1" ::"v"(a128), "v"(v4) : "memory");
        if (MODE == R2X32_ALL) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:64" : "=v"(v2) : "v"(a32) : "memory");
        if (MODE == R32_ALL) asm volatile("ds_read_b32 %0, %1" : "=v"(acc) : "v"(a32) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + v4.x + v4.w + v2.x + w2.y;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
double run(float* out, long long* cycles, int cus) {
  const int blocks = cus * 2;                       // two blocks of 16 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL((lds_loop<MODE>), dim3(blocks), dim3(1024), 0, 0, out, cycles);
  hipLaunchKernelGGL((lds_loop<MODE>), dim3(blocks), dim3(1024), 0, 0, out, cycles);
  if (hipDeviceSynchronize() != hipSuccess) return -1.0;
  std::vector<long long> h(blocks);
  if (hipMemcpy(h.data(), cycles, blocks * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
  double sum = 0;
  for (long long c : h) sum += (double)c;
  // 32 waves per CU, each kIters * kPer instructions, in the mean time a block took
  return sum / blocks / ((double)kIters * kPer * 32);
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float* out;
  long long* cycles;
  CK(hipMalloc(&out, (size_t)cus * 2 * 1024 * 4));
  CK(hipMalloc(&cycles, (size_t)cus * 2 * sizeof(long long)));
  printf("%s: %d CUs; clock64 ticks of the CU per wave-instruction, 32 waves per CU\n", p.gcnArchName, cus);
#define ROW(MODE, name) printf("%-46s %8.2f\n", name, run<MODE>(out, cycles, cus));
  ROW(W32_ALL, "ds_write_b32, 64 lanes")
  ROW(W32_LOW_HALF, "ds_write_b32, lanes 0-31")
  ROW(W32_EVEN, "ds_write_b32, even lanes")
  ROW(W32_ONE, "ds_write_b32, lane 0")
  ROW(ADDTID_ALL, "ds_write_addtid_b32, 64 lanes (+ s_mov m0)")
  ROW(W2X32_ALL, "ds_write2_b32 (two rows), 64 lanes")
  ROW(W64_ALL, "ds_write_b64, 64 lanes")
  ROW(W128_ALL, "ds_write_b128, 64 lanes")
  ROW(R32_ALL, "ds_read_b32, 64 lanes")
  ROW(R2X32_ALL, "ds_read2_b32 (two rows), 64 lanes")
  ROW(R128_ALL, "ds_read_b128, 64 lanes")
  ROW(R128_LOW_HALF, "ds_read_b128, lanes 0-31")
  ROW(R64_BCAST_ALL, "ds_read_b64 of one address, 64 lanes")
  ROW(R64_BCAST_ONE, "ds_read_b64 of one address, lane 0")
  ROW(R2X64_BCAST_ALL, "ds_read2_b64 of one address, 64 lanes")
  fflush(stdout);
  return 0;
}
