// Photometric loss of the training step, fused for gfx950:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean SSIM(x, y))
// (reference: train.py:91-92 with utils/loss_utils.py:17-63 — 11x11 Gaussian
// window, sigma 1.5, zero padding, per channel, C1 = 0.01^2, C2 = 0.03^2).
//
// The reference evaluates SSIM with five grouped 11x11 convolutions and a dozen
// elementwise kernels over (3,H,W); at 1080p that is several milliseconds of
// library convolutions around a 25 MB image.  Here one kernel per direction
// does it from LDS: a 16x16 output tile loads its 26x26 halo of both images
// once, filters the five moments separably (horizontal, then vertical), and
// forms the SSIM value plus the three partial derivatives the backward needs;
// the backward filters those three maps the same way.  Both kernels are
// HBM-bound: forward reads 8 B and writes 12 B per pixel-channel, backward
// reads 20 B and writes 4 B.
#include "gsr_common.h"

namespace {

constexpr int kTile = 16;
constexpr int kR = 5;                  // window radius (11 taps)
constexpr int kIn = kTile + 2 * kR;    // 26

struct Window { float w[2 * kR + 1]; };

// Normalised 1-D Gaussian, sigma 1.5 (utils/loss_utils.py:23-25), computed on
// the host in double and rounded to fp32.
Window make_window() {
  Window g;
  double s = 0.0, v[2 * kR + 1];
  for (int i = 0; i <= 2 * kR; i++) { v[i] = exp(-double((i - kR) * (i - kR)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
  for (int i = 0; i <= 2 * kR; i++) g.w[i] = (float)(v[i] / s);
  return g;
}

constexpr float kC1 = 0.01f * 0.01f;
constexpr float kC2 = 0.03f * 0.03f;

// block = 256 threads, grid = (tiles_x, tiles_y, channels)
__global__ void __launch_bounds__(256) loss_fwd_kernel(int H, int W, const float* __restrict__ img,
                                                       const float* __restrict__ gt, Window win,
                                                       float* __restrict__ dmaps, float* __restrict__ partial) {
  __shared__ float sx[kIn][kIn + 1], sy[kIn][kIn + 1];
  __shared__ float hx[kIn][kTile], hy[kIn][kTile], hxx[kIn][kTile], hyy[kIn][kTile], hxy[kIn][kTile];
  __shared__ float red[2][4];
  const int c = blockIdx.z;
  const size_t plane = (size_t)H * W;
  const float* X = img + c * plane;
  const float* Y = gt + c * plane;
  const int x0 = blockIdx.x * kTile - kR, y0 = blockIdx.y * kTile - kR;
  for (int i = threadIdx.x; i < kIn * kIn; i += 256) {
    const int r = i / kIn, q = i % kIn, gy = y0 + r, gx = x0 + q;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    sx[r][q] = in ? X[(size_t)gy * W + gx] : 0.f;
    sy[r][q] = in ? Y[(size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kIn * kTile; i += 256) {
    const int r = i / kTile, q = i % kTile;
    float ax = 0.f, ay = 0.f, axx = 0.f, ayy = 0.f, axy = 0.f;
#pragma unroll
    for (int k = 0; k <= 2 * kR; k++) {
      const float a = sx[r][q + k], b = sy[r][q + k], w = win.w[k];
      ax += w * a; ay += w * b; axx += w * (a * a); ayy += w * (b * b); axy += w * (a * b);
    }
    hx[r][q] = ax; hy[r][q] = ay; hxx[r][q] = axx; hyy[r][q] = ayy; hxy[r][q] = axy;
  }
  __syncthreads();
  const int lx = threadIdx.x % kTile, ly = threadIdx.x / kTile;
  const int gx = blockIdx.x * kTile + lx, gy = blockIdx.y * kTile + ly;
  float l1 = 0.f, ss = 0.f;
  if (gx < W && gy < H) {
    float mu1 = 0.f, mu2 = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
    for (int k = 0; k <= 2 * kR; k++) {
      const float w = win.w[k];
      mu1 += w * hx[ly + k][lx]; mu2 += w * hy[ly + k][lx];
      exx += w * hxx[ly + k][lx]; eyy += w * hyy[ly + k][lx]; exy += w * hxy[ly + k][lx];
    }
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = exx - mu1_sq, s2 = eyy - mu2_sq, s12 = exy - mu12;
    const float A = mu1_sq + mu2_sq + kC1, B = s1 + s2 + kC2, C = 2.0f * mu12 + kC1, D = 2.0f * s12 + kC2;
    const float invA = 1.0f / A, invB = 1.0f / B;
    const float S = C * D * invA * invB;
    // partial derivatives of S w.r.t. (sigma1^2, sigma12) and the total derivative w.r.t. mu1
    // with E[x^2], E[xy] held fixed (sigma1^2 = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2)
    const float dS_ds1 = -S * invB;
    const float dS_ds12 = 2.0f * C * invA * invB;
    const float dS_dmu1 = 2.0f * mu2 * D * invA * invB - 2.0f * mu1 * S * invA - 2.0f * mu1 * dS_ds1 - mu2 * dS_ds12;
    const size_t o = c * plane + (size_t)gy * W + gx, cs = 3 * plane;   // dmaps: (3 maps, C, H, W) with C = 3
    dmaps[o] = dS_dmu1;
    dmaps[cs + o] = dS_ds1;
    dmaps[2 * cs + o] = dS_ds12;
    ss = S;
    l1 = fabsf(sx[ly + kR][lx + kR] - sy[ly + kR][lx + kR]);
  }
  // block sums -> one (l1, ssim) pair per block, summed later in a fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { l1 += __shfl_xor(l1, o); ss += __shfl_xor(ss, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = l1; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// out = (loss, l1 mean, ssim mean); one block, fixed summation order, double accumulation.
__global__ void __launch_bounds__(256) loss_finalize_kernel(int nblocks, const float* __restrict__ partial, double n,
                                                            float lambda, float* __restrict__ out) {
  __shared__ double r0[256], r1[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += (double)partial[2 * i]; b += (double)partial[2 * i + 1]; }
  r0[threadIdx.x] = a; r1[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double l1 = r0[0] / n, ss = r1[0] / n;
    out[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ss));
    out[1] = (float)l1;
    out[2] = (float)ss;
  }
}

__global__ void __launch_bounds__(256) loss_bwd_kernel(int H, int W, const float* __restrict__ img,
                                                       const float* __restrict__ gt, const float* __restrict__ dmaps,
                                                       Window win, const float* __restrict__ grad_out, float lambda,
                                                       float inv_n, float* __restrict__ grad_img) {
  __shared__ float sm[3][kIn][kIn + 1];
  __shared__ float hm[3][kIn][kTile];
  const int c = blockIdx.z;
  const size_t plane = (size_t)H * W, cs = 3 * plane;
  const int x0 = blockIdx.x * kTile - kR, y0 = blockIdx.y * kTile - kR;
  for (int i = threadIdx.x; i < kIn * kIn; i += 256) {
    const int r = i / kIn, q = i % kIn, gy = y0 + r, gx = x0 + q;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = c * plane + (size_t)gy * W + gx;
    sm[0][r][q] = in ? dmaps[o] : 0.f;
    sm[1][r][q] = in ? dmaps[cs + o] : 0.f;
    sm[2][r][q] = in ? dmaps[2 * cs + o] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kIn * kTile; i += 256) {
    const int r = i / kTile, q = i % kTile;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k <= 2 * kR; k++) {
      const float w = win.w[k];
      a0 += w * sm[0][r][q + k]; a1 += w * sm[1][r][q + k]; a2 += w * sm[2][r][q + k];
    }
    hm[0][r][q] = a0; hm[1][r][q] = a1; hm[2][r][q] = a2;
  }
  __syncthreads();
  const int lx = threadIdx.x % kTile, ly = threadIdx.x / kTile;
  const int gx = blockIdx.x * kTile + lx, gy = blockIdx.y * kTile + ly;
  if (gx >= W || gy >= H) return;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int k = 0; k <= 2 * kR; k++) {
    const float w = win.w[k];
    c0 += w * hm[0][ly + k][lx]; c1 += w * hm[1][ly + k][lx]; c2 += w * hm[2][ly + k][lx];
  }
  const size_t o = c * plane + (size_t)gy * W + gx;
  const float x = img[o], y = gt[o];
  const float go = grad_out[0];
  const float d = x - y;
  const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
  const float dssim = c0 + 2.0f * x * c1 + y * c2;     // d(sum of SSIM map) / dx
  grad_img[o] = go * inv_n * ((1.0f - lambda) * sgn - lambda * dssim);
}

}  // namespace

extern "C" size_t gsr_loss_partial_count(int C, int H, int W) {
  return (size_t)C * gsr::div_up(H, kTile) * gsr::div_up(W, kTile);
}

extern "C" int gsr_loss_fwd(int C, int H, int W, const float* img, const float* gt, float lambda_dssim, float* dmaps,
                            float* partial, float* out3, void* stream) {
  GSR_REQUIRE(C == 3 && H > 0 && W > 0, "loss_fwd: expects (3,H,W), got C=%d H=%d W=%d", C, H, W);
  GSR_REQUIRE(img && gt && dmaps && partial && out3, "loss_fwd: null pointer");
  static const Window win = make_window();
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(gsr::div_up(W, kTile), gsr::div_up(H, kTile), C);
  hipLaunchKernelGGL(loss_fwd_kernel, grid, dim3(256), 0, st, H, W, img, gt, win, dmaps, partial);
  int rc = gsr::check_launch("loss_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (int)(grid.x * grid.y * grid.z), partial,
                     (double)C * H * W, lambda_dssim, out3);
  return gsr::check_launch("loss_fwd: finalize");
}

extern "C" int gsr_loss_bwd(int C, int H, int W, const float* img, const float* gt, const float* dmaps,
                            const float* grad_out, float lambda_dssim, float* grad_img, void* stream) {
  GSR_REQUIRE(C == 3 && H > 0 && W > 0, "loss_bwd: expects (3,H,W), got C=%d H=%d W=%d", C, H, W);
  GSR_REQUIRE(img && gt && dmaps && grad_out && grad_img, "loss_bwd: null pointer");
  static const Window win = make_window();
  dim3 grid(gsr::div_up(W, kTile), gsr::div_up(H, kTile), C);
  hipLaunchKernelGGL(loss_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, img, gt, dmaps, win, grad_out,
                     lambda_dssim, 1.0f / ((float)C * (float)H * (float)W), grad_img);
  return gsr::check_launch("loss_bwd");
}
