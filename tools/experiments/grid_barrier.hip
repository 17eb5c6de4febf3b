// What a device-wide barrier costs inside one resident grid on this part (8 XCDs, each with its own L2):
// every workgroup adds one to a counter in device memory and polls it until all have arrived.  Bounded
// polling: a workgroup that does not see the others within kSpinLimit polls sets a flag and leaves, and every
// later barrier returns at once, so the kernel always ends.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned* failed, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ++spins > kSpinLimit) {
        __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

// Two levels: a workgroup arrives at the counter of its group (block index mod 8: the workgroups of one XCD when
// the grid is dealt round-robin), the last arrival of a group arrives at the top counter, everybody polls the top
// one.  Eight chains of 1/8 of the same-address atomics in parallel, then one of eight.
__device__ __forceinline__ bool grid_barrier2(unsigned* counters, unsigned* failed, unsigned round) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned groups = gridDim.x >= 8 ? 8 : 1, g = blockIdx.x % groups;
    const unsigned members = gridDim.x / groups + (g < gridDim.x % groups ? 1 : 0);
    __threadfence();
    const unsigned before = __hip_atomic_fetch_add(counters + 32 * (1 + g), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1 == members * (round + 1))
      __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counters, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < groups * (round + 1)) {
      if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ++spins > kSpinLimit) {
        __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

__global__ void __launch_bounds__(256) barrier2_kernel(unsigned* counters, unsigned* failed, unsigned* data, int rounds) {
  for (int r = 0; r < rounds; r++) {
    data[(size_t)blockIdx.x * 256 + threadIdx.x] += (unsigned)r;
    if (!grid_barrier2(counters, failed, (unsigned)r)) return;
  }
}

// `rounds` barriers in a row; between two barriers every thread touches one word of `data` (so that the barrier
// also has to make ordinary stores visible, as a real phase boundary would).
__global__ void __launch_bounds__(256) barrier_kernel(unsigned* counter, unsigned* failed, unsigned* data, int rounds) {
  const unsigned n = gridDim.x;
  for (int r = 0; r < rounds; r++) {
    data[(size_t)blockIdx.x * 256 + threadIdx.x] += (unsigned)r;
    if (!grid_barrier(counter, failed, n * (unsigned)(r + 1))) return;
  }
}

__global__ void __launch_bounds__(256) empty_kernel(unsigned* data) { data[(size_t)blockIdx.x * 256 + threadIdx.x] += 1u; }

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  unsigned *counter, *failed, *data;
  hipMalloc(&counter, 4 * 32 * 9);
  hipMalloc(&failed, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int per_cu : {1, 2, 4}) {
    const int blocks = cus * per_cu;
    int fit = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, (const void*)barrier_kernel, 256, 0);
    if (fit < per_cu) { printf("%d blocks per CU do not fit (%d do)\n", per_cu, fit); continue; }
    hipMalloc(&data, (size_t)blocks * 256 * 4);
    hipMemset(data, 0, (size_t)blocks * 256 * 4);
    for (int rounds : {1, 11, 101}) {
      float best = 1e30f;
      unsigned bad = 0;
      for (int rep = 0; rep < 5; rep++) {
        hipMemset(counter, 0, 4);
        hipMemset(failed, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(barrier_kernel, dim3(blocks), dim3(256), 0, 0, counter, failed, data, rounds);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        unsigned f = 0;
        hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
        bad |= f;
      }
      printf("%4d workgroups (%d per CU), %3d barriers: %.1f us%s\n", blocks, per_cu, rounds, best * 1e3, bad ? "  (a barrier gave up)" : "");
      best = 1e30f;
      bad = 0;
      for (int rep = 0; rep < 5; rep++) {
        hipMemset(counter, 0, 4 * 32 * 9);
        hipMemset(failed, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(barrier2_kernel, dim3(blocks), dim3(256), 0, 0, counter, failed, data, rounds);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        unsigned f = 0;
        hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
        bad |= f;
      }
      printf("%4d workgroups (%d per CU), %3d two-level barriers: %.1f us%s\n", blocks, per_cu, rounds, best * 1e3, bad ? "  (a barrier gave up)" : "");
    }
    // slope between 11 and 101 barriers = one barrier
    hipFree(data);
  }
  // for comparison: back-to-back launches of an empty kernel of the same grid
  hipMalloc(&data, (size_t)cus * 4 * 256 * 4);
  for (int n : {1, 11, 101}) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      for (int i = 0; i < n; i++) hipLaunchKernelGGL(empty_kernel, dim3(cus), dim3(256), 0, 0, data);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%3d launches of a %d-workgroup kernel back to back: %.1f us\n", n, cus, best * 1e3);
  }
  return 0;
}
