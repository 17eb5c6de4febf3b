// Fused Adam step for the Gaussian model's parameter tensors on gfx950.
//
// The reference optimises six tensors with torch.optim.Adam(lr per group,
// eps = 1e-15, betas (0.9, 0.999), no weight decay, no amsgrad)
// (scene/gaussian_model.py:154-163, train.py:126-128).  PyTorch runs that as a
// chain of elementwise passes per tensor; here one launch updates all tensors:
// every element is read once (parameter, gradient, both moments: 16 B) and
// written once (12 B), which is the HBM minimum of 28 B per element.
//
// Arithmetic follows torch.optim.Adam's single-tensor path operation by
// operation:
//     m <- m + (1 - b1) (g - m)                      (lerp)
//     v <- b2 v + (1 - b2) g g                       (mul, addcmul)
//     p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with the bias corrections and the weights 1 - b1, 1 - b2 formed on the host in double, as
// Python does (the hyper-parameters cross the ABI as doubles for that reason).
// `grad_scale` multiplies the gradient first (1/world_size after a summing
// all-reduce; 1 otherwise).
#include "gsr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 4;                    // one 16-byte access per array
// 16-byte accesses per array and thread, a block's width apart, all of a thread's loads ahead of its arithmetic; and
// every access nontemporal: each element is touched once per step, so nothing is gained by keeping it in the L2.
// 1M Gaussians (59M elements): one round with ordinary accesses 332 us, nontemporal 317, two rounds 304, three 299,
// four 302.
constexpr int kRounds = 3;
constexpr int kPerBlock = kBlock * kPerThread * kRounds;
#define GSR_LD(p) __builtin_nontemporal_load(p)
#define GSR_ST(p, v) __builtin_nontemporal_store(v, p)

struct AdamTensors {
  GsrAdamTensor t[GSR_ADAM_MAX_TENSORS];
  uint32_t first_block[GSR_ADAM_MAX_TENSORS + 1];
  int count;
};

// w1 = 1 - b1 and w2 = 1 - b2 are formed on the host in double and rounded once, as Python forms them
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float w1, float b2, float w2,
                                            float step_size, float inv_bc2_sqrt, float eps, float grad_scale) {
  g = g * grad_scale;
  m = m + w1 * (g - m);
  v = v * b2 + w2 * g * g;
  const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(kBlock) adam_kernel(AdamTensors a, float w1, float b2, float w2, float eps, float bc1,
                                                      float inv_bc2_sqrt, float grad_scale) {
  // which tensor this block works on: at most GSR_ADAM_MAX_TENSORS, linear search in scalar registers
  int k = 0;
  while (k + 1 < a.count && blockIdx.x >= a.first_block[k + 1]) k++;
  const GsrAdamTensor t = a.t[k];
  const size_t block_base = (size_t)(blockIdx.x - a.first_block[k]) * kPerBlock + (size_t)threadIdx.x * kPerThread;
  const float step_size = t.lr / bc1;
  if (t.vec4 && block_base - (size_t)threadIdx.x * kPerThread + kPerBlock <= t.n) {
    // a whole block of vector accesses: all loads of the thread's rounds first
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 p[kRounds], g[kRounds], m[kRounds], v[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
      const size_t base = block_base + (size_t)r * kBlock * kPerThread;
      p[r] = GSR_LD(reinterpret_cast<f4*>(t.param + base));
      g[r] = GSR_LD(reinterpret_cast<const f4*>(t.grad + base));
      m[r] = GSR_LD(reinterpret_cast<f4*>(t.exp_avg + base));
      v[r] = GSR_LD(reinterpret_cast<f4*>(t.exp_avg_sq + base));
    }
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
      const size_t base = block_base + (size_t)r * kBlock * kPerThread;
      float px = p[r].x, py = p[r].y, pz = p[r].z, pw = p[r].w, mx = m[r].x, my = m[r].y, mz = m[r].z, mw = m[r].w;
      float vx = v[r].x, vy = v[r].y, vz = v[r].z, vw = v[r].w;
      adam_update(px, g[r].x, mx, vx, w1, b2, w2, step_size, inv_bc2_sqrt, eps, grad_scale);
      adam_update(py, g[r].y, my, vy, w1, b2, w2, step_size, inv_bc2_sqrt, eps, grad_scale);
      adam_update(pz, g[r].z, mz, vz, w1, b2, w2, step_size, inv_bc2_sqrt, eps, grad_scale);
      adam_update(pw, g[r].w, mw, vw, w1, b2, w2, step_size, inv_bc2_sqrt, eps, grad_scale);
      const f4 po = {px, py, pz, pw}, mo = {mx, my, mz, mw}, vo = {vx, vy, vz, vw};
      GSR_ST(reinterpret_cast<f4*>(t.param + base), po);
      GSR_ST(reinterpret_cast<f4*>(t.exp_avg + base), mo);
      GSR_ST(reinterpret_cast<f4*>(t.exp_avg_sq + base), vo);
    }
    return;
  }
  for (int r = 0; r < kRounds; r++) {
    const size_t base = block_base + (size_t)r * kBlock * kPerThread;
    for (size_t i = base; i < base + kPerThread && i < t.n; i++) {
      float p = t.param[i], m = t.exp_avg[i], v = t.exp_avg_sq[i];
      adam_update(p, t.grad[i], m, v, w1, b2, w2, step_size, inv_bc2_sqrt, eps, grad_scale);
      t.param[i] = p; t.exp_avg[i] = m; t.exp_avg_sq[i] = v;
    }
  }
}

}  // namespace

extern "C" int gsr_adam_step(int count, const GsrAdamTensor* h_tensors, double beta1, double beta2, double eps, int step,
                             float grad_scale, void* stream) {
  GSR_REQUIRE(count >= 0 && count <= GSR_ADAM_MAX_TENSORS, "adam_step: %d tensors (at most %d)", count,
              GSR_ADAM_MAX_TENSORS);
  GSR_REQUIRE(step >= 1, "adam_step: step counts from 1, got %d", step);
  GSR_REQUIRE(count == 0 || h_tensors, "adam_step: null tensor list");
  AdamTensors a{};
  a.count = 0;
  uint32_t blocks = 0;
  for (int k = 0; k < count; k++) {
    const GsrAdamTensor& t = h_tensors[k];
    if (t.n == 0) continue;
    GSR_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, "adam_step: null pointer in tensor %d", k);
    GsrAdamTensor u = t;
    u.vec4 = (((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0;
    a.t[a.count] = u;
    a.first_block[a.count] = blocks;
    blocks += (uint32_t)((t.n + kPerBlock - 1) / kPerBlock);
    a.count++;
  }
  a.first_block[a.count] = blocks;
  if (blocks == 0) return GSR_OK;
  // bias corrections as Python forms them (double), then rounded once
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2_sqrt = sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, a, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, (float)bc1, (float)(1.0 / bc2_sqrt), grad_scale);
  return gsr::check_launch("adam_step");
}
