// Depth-sort microbenchmark: (u32 key, u64 value) pairs, public rocPRIM radix sort against a
// driver of its Onesweep kernels that clears all per-pass state in one place, over block shapes.
//   hipcc --offload-arch=gfx950 -O3 sort_bench.hip -o sort_bench && ./sort_bench [P]
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdarg>
#include "../../gaussian-splatting-own_amd/csrc/depthsort.hip"
namespace gsr { int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); return code; } }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
namespace rp = rocprim;
namespace rd = rocprim::detail;
using Bid = rd::block_id_wrapper<unsigned int, true>;

template <unsigned HB, unsigned HI>
__global__ void __launch_bounds__(HB) hist_kernel(const uint32_t* keys, uint32_t* counts, uint32_t size,
                                                  uint32_t full_blocks, uint32_t* zero_from, uint32_t zero_words) {
  for (uint32_t i = blockIdx.x * HB + threadIdx.x; i < zero_words; i += gridDim.x * HB) zero_from[i] = 0u;
  rd::onesweep_histograms<HB, HI, 8, false>(keys, counts, size, full_blocks, rp::identity_decomposer{}, 0u, 32u);
}
template <unsigned HB>
__global__ void __launch_bounds__(HB) scan_kernel(uint32_t* counts) { rd::onesweep_scan_histograms<HB, 8>(counts); }

template <unsigned SB, unsigned SI, rp::block_radix_rank_algorithm A>
__global__ void __launch_bounds__(SB) pass_kernel(const uint32_t* kin, uint32_t* kout, const u64* vin, u64* vout,
                                                  uint32_t size, uint32_t* offs_in, uint32_t* offs_out,
                                                  rd::onesweep_lookback_state* lb, uint32_t bit, uint32_t full_blocks,
                                                  Bid bid) {
  rd::onesweep_iteration<SB, SI, 8, false, A>(kin, kout, vin, vout, size, offs_in, offs_out, lb,
                                              rp::identity_decomposer{}, bit, 8u, full_blocks, bid);
}

template <unsigned HB, unsigned HI, unsigned SB, unsigned SI, rp::block_radix_rank_algorithm A>
struct Own {
  static size_t bytes(uint32_t P) {
    const uint32_t blocks = (P + SB * SI - 1) / (SB * SI);
    return 4096 + 1024 + 256 + (size_t)4 * 256 * blocks * 4 + (size_t)P * 12;
  }
  static void run(uint32_t P, const uint32_t* keys, uint32_t* keys_out, const u64* vals, u64* vals_out, char* tmp,
                  hipStream_t st) {
    const uint32_t blocks = (P + SB * SI - 1) / (SB * SI), full = P % (SB * SI) == 0 ? blocks : blocks - 1;
    const uint32_t hblocks = (P + HB * HI - 1) / (HB * HI), hfull = P % (HB * HI) == 0 ? hblocks : hblocks - 1;
    uint32_t* counts = (uint32_t*)tmp;                 // 4 x 256
    uint32_t* offs_tmp = (uint32_t*)(tmp + 4096);      // 256
    uint32_t* bids = (uint32_t*)(tmp + 4096 + 1024);   // 4 counters (+ lookback states behind them)
    rd::onesweep_lookback_state* lb = (rd::onesweep_lookback_state*)(tmp + 4096 + 1024 + 256);
    const size_t lb_words = (size_t)4 * 256 * blocks;
    uint32_t* ktmp = (uint32_t*)(tmp + 4096 + 1024 + 256 + lb_words * 4);
    u64* vtmp = (u64*)((char*)ktmp + (((size_t)P * 4 + 255) & ~(size_t)255));
    (void)hipMemsetAsync(counts, 0, 4096, st);
    hipLaunchKernelGGL((hist_kernel<HB, HI>), dim3(hblocks), dim3(HB), 0, st, keys, counts, P, hfull, bids,
                       (uint32_t)(64 + lb_words));
    hipLaunchKernelGGL((scan_kernel<HB>), dim3(4), dim3(HB), 0, st, counts);
    const uint32_t* kin = keys; const u64* vin = vals;
    for (int pass = 0; pass < 4; pass++) {
      uint32_t* ko = (pass & 1) ? keys_out : ktmp;
      u64* vo = (pass & 1) ? vals_out : vtmp;
      hipLaunchKernelGGL((pass_kernel<SB, SI, A>), dim3(blocks), dim3(SB), 0, st, kin, ko, vin, vo, P, counts + 256 * pass,
                         offs_tmp, lb + (size_t)pass * 256 * blocks, 8u * pass, full, Bid::create(bids + pass));
      kin = ko; vin = vo;
    }
  }
};

template <class F>
float time_ms(F f, int reps = 30) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; i++) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

template <unsigned HB, unsigned HI, unsigned SB, unsigned SI, rp::block_radix_rank_algorithm A>
void try_own(const char* name, uint32_t P, const uint32_t* keys, uint32_t* ko, const u64* vals, u64* vo,
             const std::vector<uint32_t>& want_k, const std::vector<u64>& want_v) {
  using S = Own<HB, HI, SB, SI, A>;
  char* tmp; CK(hipMalloc(&tmp, S::bytes(P) + 4096));
  CK(hipMemset(ko, 0, (size_t)P * 4)); CK(hipMemset(vo, 0, (size_t)P * 8));
  S::run(P, keys, ko, vals, vo, tmp, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> k(P); std::vector<u64> v(P);
  CK(hipMemcpy(k.data(), ko, (size_t)P * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(v.data(), vo, (size_t)P * 8, hipMemcpyDeviceToHost));
  const bool ok = k == want_k && v == want_v;
  const float ms = time_ms([&] { S::run(P, keys, ko, vals, vo, tmp, 0); });
  printf("own  %-28s %8.1f us  %s\n", name, ms * 1e3f, ok ? "exact" : "MISMATCH");
  fflush(stdout);
  CK(hipFree(tmp));
}

void try_handwritten(uint32_t P, const uint32_t* keys, const u64* vals, u64* vo, const std::vector<uint32_t>& want_k,
                     const std::vector<u64>& want_v) {
  const size_t bytes = gsr::depth_sort_bytes((int)P);
  char* tmp; CK(hipMalloc(&tmp, bytes));
  CK(hipMemset(vo, 0xEE, (size_t)P * 8));
  gsr::depth_sort((int)P, keys, (const uint64_t*)vals, (uint64_t*)vo, tmp, 0);
  CK(hipDeviceSynchronize());
  uint32_t V = 0;
  while (V < P && want_k[V] != 0xFFFFFFFFu) V++;
  std::vector<u64> v(P);
  CK(hipMemcpy(v.data(), vo, (size_t)P * 8, hipMemcpyDeviceToHost));
  bool ok = true;
  for (uint32_t i = 0; i < V && ok; i++) ok = v[i] == want_v[i];
  for (uint32_t i = V; i < P && ok; i++) ok = v[i] == 0xEEEEEEEEEEEEEEEEull;   // untouched beyond the visible ones
  uint32_t ctl[16];
  CK(hipMemcpy(ctl, tmp + 4096, sizeof(ctl), hipMemcpyDeviceToHost));
  const float ms = time_ms([&] { gsr::depth_sort((int)P, keys, (const uint64_t*)vals, (uint64_t*)vo, tmp, 0); });
#ifdef GSR_SORT_TIMING
  {
    static long long t[4][512][8];
    CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_sort_times), sizeof(t)));
    const int nb = (int)((P + 4095) / 4096) / (P >= 4500000u ? 2 : 1);
    for (int pass = 0; pass < 4; pass++) {
      long long t0 = t[pass][0][0];
      for (int b = 0; b < nb && b < 512; b++) t0 = t[pass][b][0] < t0 ? t[pass][b][0] : t0;
      for (int b : {0, 1, 15, 16, 100, 200, nb - 1}) {
        if (b >= nb || b >= 512) continue;
        printf("pass %d wg %3d start %6.2f us | load %5.2f rank %5.2f sums %5.2f lookback %5.2f place %5.2f write %5.2f\n", pass, b,
               (t[pass][b][0] - t0) * 0.01, (t[pass][b][1] - t[pass][b][0]) * 0.01, (t[pass][b][2] - t[pass][b][1]) * 0.01,
               (t[pass][b][3] - t[pass][b][2]) * 0.01, (t[pass][b][4] - t[pass][b][3]) * 0.01,
               (t[pass][b][5] - t[pass][b][4]) * 0.01, (t[pass][b][6] - t[pass][b][5]) * 0.01);
      }
    }
  }
#endif
  printf("hand-written (visible %u, counted %u, error word %u) %8.1f us  %s\n", V, ctl[8] + ctl[9] + ctl[10] + ctl[11] + ctl[12] + ctl[13] + ctl[14] + ctl[15], ctl[4], ms * 1e3f,
         ok ? "exact" : "MISMATCH");
  fflush(stdout);
  CK(hipFree(tmp));
}

template <class Cfg>
void try_public(const char* name, uint32_t P, const uint32_t* keys, uint32_t* ko, const u64* vals, u64* vo) {
  size_t bytes = 0;
  CK((rp::radix_sort_pairs<Cfg>(nullptr, bytes, keys, ko, vals, vo, (size_t)P, 0, 32)));
  void* tmp; CK(hipMalloc(&tmp, bytes));
  const float ms = time_ms([&] { (void)rp::radix_sort_pairs<Cfg>(tmp, bytes, keys, ko, vals, vo, (size_t)P, 0, 32, 0); });
  printf("rocprim %-25s %8.1f us\n", name, ms * 1e3f);
  fflush(stdout);
  CK(hipFree(tmp));
}

int main(int argc, char** argv) {
  const uint32_t P = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u;
  std::vector<uint32_t> hk(P); std::vector<u64> hv(P);
  uint32_t s = 12345u;
  for (uint32_t i = 0; i < P; i++) {
    s = s * 1664525u + 1013904223u;
    const float z = 0.2f + 9.8f * (float)(s >> 8) / 16777216.0f;
    uint32_t bits; memcpy(&bits, &z, 4);
    hk[i] = (s % 5u == 0u) ? 0xFFFFFFFFu : bits;        // one in five culled
    hv[i] = ((u64)(s >> 3) << 24) | i;
  }
  std::vector<uint32_t> order(P);
  for (uint32_t i = 0; i < P; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hk[a] < hk[b]; });
  std::vector<uint32_t> wk(P); std::vector<u64> wv(P);
  for (uint32_t i = 0; i < P; i++) { wk[i] = hk[order[i]]; wv[i] = hv[order[i]]; }
  uint32_t *keys, *ko; u64 *vals, *vo;
  CK(hipMalloc(&keys, (size_t)P * 4)); CK(hipMalloc(&ko, (size_t)P * 4));
  CK(hipMalloc(&vals, (size_t)P * 8)); CK(hipMalloc(&vo, (size_t)P * 8));
  CK(hipMemcpy(keys, hk.data(), (size_t)P * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vals, hv.data(), (size_t)P * 8, hipMemcpyHostToDevice));
  printf("P = %u\n", P);
  using Dflt = rp::radix_sort_config<rp::default_config, rp::default_config, rp::default_config, (1u << 16)>;
  try_public<Dflt>("default", P, keys, ko, vals, vo);
  constexpr auto M = rp::block_radix_rank_algorithm::match;
  try_own<1024, 4, 1024, 4, M>("library kernels, 1024x4", P, keys, ko, vals, vo, wk, wv);
  try_handwritten(P, keys, vals, vo, wk, wv);
  return 0;
}
