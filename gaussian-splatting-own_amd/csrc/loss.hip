// Photometric loss of the training step, fused for gfx950:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean SSIM(x, y))
// (reference: train.py:91-92 with utils/loss_utils.py:17-63 — 11x11 Gaussian
// window, sigma 1.5, zero padding, per channel, C1 = 0.01^2, C2 = 0.03^2).
//
// The reference evaluates SSIM with five grouped 11x11 convolutions and a dozen
// elementwise kernels over (3,H,W); at 1080p that is several milliseconds of
// library convolutions around a 25 MB image.  Here one kernel per direction
// does it on 32x32 output tiles with their 42x42 halo, separably (horizontal
// pass into LDS, then vertical).  The forward filters four moments (the two
// means, E[x^2] + E[y^2] (SSIM uses the variances only as their sum) and
// E[xy]) and forms the SSIM value plus the three partial derivatives the
// backward needs; the backward filters those three maps the same way.  Both
// filter passes are register-blocked: a thread produces four consecutive
// outputs along the filter direction from fourteen inputs it holds in
// registers, so that an output costs 3.5 reads per moment instead of 11.
// The forward's horizontal pass reads its inputs straight from global memory
// (14 consecutive floats per task; the overlap between tasks is the L1's
// business): staging the halo of both images in LDS first cost a barrier and
// 15 KB per workgroup, and the kernel is bound by how well the workgroups of a
// CU overlap one's arithmetic with another's loads and barriers (0.061 -> 0.050
// ms at 1080p).  The backward keeps the staged halo with the next tile's loads
// prefetched into registers: the direct form is no faster there (three maps,
// hence three load latencies in a row or too many registers).  HBM traffic:
// forward reads 8 B and writes 12 B per pixel-channel, backward reads 20 B and
// writes 4 B.
#include "gsr_common.h"

namespace {

constexpr int kR = 5;                    // window radius (11 taps)
constexpr int kTaps = 2 * kR + 1;
constexpr int kTX = 32, kTY = 32;        // output tile
constexpr int kIW = kTX + 2 * kR;        // 42 input columns
constexpr int kIH = kTY + 2 * kR;        // 42 input rows
constexpr int kIS = 44;                  // LDS row stride of a staged input tile: a multiple of 4 floats
constexpr int kOut = 4;                  // outputs per thread along the filter direction
constexpr int kNeed = kOut + 2 * kR;     // 14 inputs for them
constexpr int kHTasks = kIH * (kTX / kOut);          // horizontal pass: 42 rows x 8 column groups
constexpr int kVTasks = kTX * (kTY / kOut);          // vertical pass: 32 columns x 8 row groups
constexpr int kLoads = (kIH * kIW + 255) / 256;      // halo elements per thread
static_assert(kVTasks == 256, "the vertical pass gives every thread of the block exactly one task");

struct Window { float w[kTaps]; };

// Normalised 1-D Gaussian, sigma 1.5 (utils/loss_utils.py:23-25), computed on
// the host in double and rounded to fp32.
Window make_window() {
  Window g;
  double s = 0.0, v[kTaps];
  for (int i = 0; i < kTaps; i++) { v[i] = exp(-double((i - kR) * (i - kR)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
  for (int i = 0; i < kTaps; i++) g.w[i] = (float)(v[i] / s);
  return g;
}

constexpr float kC1 = 0.01f * 0.01f;
constexpr float kC2 = 0.03f * 0.03f;

// 14 consecutive floats of row gy of an (H,W) plane starting at column gx0, zero outside the image.
// Any 4-byte alignment: the vector type says so, and global loads of 16 bytes need no more.
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ void load14(const float* __restrict__ plane, int H, int W, int gy, int gx0, float* a) {
  if (gy < 0 || gy >= H) {
#pragma unroll
    for (int i = 0; i < kNeed; i++) a[i] = 0.f;
  } else if (gx0 >= 0 && gx0 + kNeed <= W) {
    const float* p = plane + (size_t)gy * W + gx0;
    const float4u q0 = *reinterpret_cast<const float4u*>(p), q1 = *reinterpret_cast<const float4u*>(p + 4),
                  q2 = *reinterpret_cast<const float4u*>(p + 8);
    const float2u q3 = *reinterpret_cast<const float2u*>(p + 12);
    a[0] = q0.x; a[1] = q0.y; a[2] = q0.z; a[3] = q0.w; a[4] = q1.x; a[5] = q1.y; a[6] = q1.z; a[7] = q1.w;
    a[8] = q2.x; a[9] = q2.y; a[10] = q2.z; a[11] = q2.w; a[12] = q3.x; a[13] = q3.y;
  } else {
    const float* p = plane + (size_t)gy * W;
#pragma unroll
    for (int i = 0; i < kNeed; i++) a[i] = (gx0 + i >= 0 && gx0 + i < W) ? p[gx0 + i] : 0.f;
  }
}

// 14 consecutive floats starting at a 16-byte aligned LDS address
__device__ __forceinline__ void read14(const float* p, float* a) {
  const float4 q0 = reinterpret_cast<const float4*>(p)[0], q1 = reinterpret_cast<const float4*>(p)[1],
               q2 = reinterpret_cast<const float4*>(p)[2];
  const float2 q3 = reinterpret_cast<const float2*>(p)[6];
  a[0] = q0.x; a[1] = q0.y; a[2] = q0.z; a[3] = q0.w; a[4] = q1.x; a[5] = q1.y; a[6] = q1.z; a[7] = q1.w;
  a[8] = q2.x; a[9] = q2.y; a[10] = q2.z; a[11] = q2.w; a[12] = q3.x; a[13] = q3.y;
}

// The (kIH x kIW) halo of one (H,W) plane, zero outside the image, in two steps so that the
// global loads of the NEXT tile can be in flight while the current tile is filtered:
// fetch_halo issues the loads into registers, stash_halo puts them into an LDS tile of row stride kIS.
struct Halo { float v[kLoads]; };

__device__ __forceinline__ void fetch_halo(const float* __restrict__ src, int H, int W, int x0, int y0, Halo& h) {
#pragma unroll
  for (int it = 0; it < kLoads; it++) {
    const int i = it * 256 + (int)threadIdx.x, r = i / kIW, q = i % kIW, gy = y0 + r, gx = x0 + q;
    const bool in = i < kIH * kIW && gy >= 0 && gy < H && gx >= 0 && gx < W;
    h.v[it] = in ? src[(size_t)gy * W + gx] : 0.f;
  }
}

__device__ __forceinline__ void stash_halo(const Halo& h, float* tile) {
#pragma unroll
  for (int it = 0; it < kLoads; it++) {
    const int i = it * 256 + (int)threadIdx.x, r = i / kIW, q = i % kIW;
    if (i < kIH * kIW) tile[r * kIS + q] = h.v[it];
  }
}

// Tiles are handed out so that the blocks that share an XCD (block index mod 8) walk one
// contiguous eighth of the tile sequence: neighbouring halos then meet in that XCD's L2.
struct TileWalk {
  int tiles_x, tiles_y, first, last, step;   // this block visits first, first + step, ... < last
  __device__ __forceinline__ void decode(int t, int& c, int& bx, int& by) const {
    bx = t % tiles_x;
    by = (t / tiles_x) % tiles_y;
    c = t / (tiles_x * tiles_y);
  }
};

__device__ __forceinline__ TileWalk make_walk(int H, int W, int C) {
  TileWalk w;
  w.tiles_x = (W + kTX - 1) / kTX;
  w.tiles_y = (H + kTY - 1) / kTY;
  const int total = w.tiles_x * w.tiles_y * C;
  const int groups = gridDim.x >= 8 ? 8 : 1;
  const int per = (total + groups - 1) / groups;
  const int g = blockIdx.x % groups;
  w.step = gridDim.x / groups;                       // the host launches a multiple of 8 blocks (or fewer than 8)
  w.first = g * per + (int)blockIdx.x / groups;
  w.last = min(total, (g + 1) * per);
  return w;
}

// block = 256 threads; a fixed number of blocks walks all (tile, channel) pairs
#ifndef GSR_LOSS_FWD_PER_CU
#define GSR_LOSS_FWD_PER_CU 5
#endif
#ifndef GSR_LOSS_BWD_PER_CU
#define GSR_LOSS_BWD_PER_CU 4
#endif
__global__ void __launch_bounds__(256, GSR_LOSS_FWD_PER_CU) loss_fwd_kernel(int H, int W, int C, const float* __restrict__ img,
                                                       const float* __restrict__ gt, Window win,
                                                       float* __restrict__ dmaps, float* __restrict__ partial) {
  // x, y, xx + yy, xy after the horizontal pass
  __shared__ __align__(16) float hm[4][kIH * kTX];
  __shared__ float red[2][4];
  const size_t plane = (size_t)H * W;
  const TileWalk walk = make_walk(H, W, C);
  const int lx = threadIdx.x % kTX, ly = (threadIdx.x / kTX) * kOut;
  float l1 = 0.f, ss = 0.f;
  for (int t = walk.first; t < walk.last; t += walk.step) {
    int c, bx, by;
    walk.decode(t, c, bx, by);
    const float* const xc = img + c * plane;
    const float* const yc = gt + c * plane;
    // (the 336 tasks are a round of 256 and a round of 80, which leaves two waves idle; a quarter of the tasks to every
    // wave, 64 + 20, keeps all four busy and is slower, 57.5 -> 60.3 us at 1080p: four partly filled rounds issue more;
    // letting the second round fall to waves 2 and 3 on every other tile, so that no SIMD always has it: no change)
    // (the horizontal pass at raised wave priority: of a CU's workgroups the ones in this pass are the ones whose
    // loads are still to be asked for; forward 57.4 -> 56.9 us, backward 41.5 -> 40.0 us at 1080p)
    __builtin_amdgcn_s_setprio(2);
    for (int k = threadIdx.x; k < kHTasks; k += 256) {
      const int r = k / (kTX / kOut), q = (k % (kTX / kOut)) * kOut;
      float a[kNeed], b[kNeed], o[4][kOut];
      load14(xc, H, W, by * kTY - kR + r, bx * kTX - kR + q, a);
      load14(yc, H, W, by * kTY - kR + r, bx * kTX - kR + q, b);
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int j = 0; j < kOut; j++) o[m][j] = 0.f;
#pragma unroll
      for (int i = 0; i < kNeed; i++) {
        const float x = a[i], y = b[i], sq = fmaf(x, x, y * y), xy = x * y;
#pragma unroll
        for (int j = 0; j < kOut; j++) {
          const int kk = i - j;
          if (kk >= 0 && kk < kTaps) {
            const float w = win.w[kk];
            o[0][j] += w * x; o[1][j] += w * y; o[2][j] += w * sq; o[3][j] += w * xy;
          }
        }
      }
#pragma unroll
      for (int m = 0; m < 4; m++)
        *reinterpret_cast<float4*>(&hm[m][r * kTX + q]) = make_float4(o[m][0], o[m][1], o[m][2], o[m][3]);
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();

    // this thread's own pixels of both images, for the L1 term
    const int gx = bx * kTX + lx;
    float xs[kOut], ys[kOut];
#pragma unroll
    for (int j = 0; j < kOut; j++) {
      const int gy = by * kTY + ly + j;
      const bool in = gx < W && gy < H;
      xs[j] = in ? xc[(size_t)gy * W + gx] : 0.f;
      ys[j] = in ? yc[(size_t)gy * W + gx] : 0.f;
    }
    float acc[4][kOut];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      float v[kNeed];
#pragma unroll
      for (int i = 0; i < kNeed; i++) v[i] = hm[m][(ly + i) * kTX + lx];
#pragma unroll
      for (int j = 0; j < kOut; j++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kTaps; k++) s += win.w[k] * v[j + k];
        acc[m][j] = s;
      }
    }
#pragma unroll
    for (int j = 0; j < kOut; j++) {
      const int gy = by * kTY + ly + j;
      if (gx < W && gy < H) {
        const float mu1 = acc[0][j], mu2 = acc[1][j], esq = acc[2][j], exy = acc[3][j];   // esq = E[x^2] + E[y^2]
        const float mu_sq = mu1 * mu1 + mu2 * mu2, mu12 = mu1 * mu2;
        const float s12 = exy - mu12;
        const float A = mu_sq + kC1, B = (esq - mu_sq) + kC2, Cc = 2.0f * mu12 + kC1, D = 2.0f * s12 + kC2;
        // reciprocals to within an ulp: the hardware's estimate and one Newton step (an IEEE division is ten
        // instructions, and this map costs the kernel a third of its instructions)
        float invA = __builtin_amdgcn_rcpf(A), invB = __builtin_amdgcn_rcpf(B);
        invA = fmaf(fmaf(-A, invA, 1.0f), invA, invA);
        invB = fmaf(fmaf(-B, invB, 1.0f), invB, invB);
        const float S = Cc * D * invA * invB;
        // partial derivatives of S w.r.t. (sigma1^2, sigma12) and the total derivative w.r.t. mu1
        // with E[x^2], E[xy] held fixed (sigma1^2 = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2)
        const float dS_ds1 = -S * invB;
        const float dS_ds12 = 2.0f * Cc * invA * invB;
        const float dS_dmu1 = 2.0f * mu2 * D * invA * invB - 2.0f * mu1 * S * invA - 2.0f * mu1 * dS_ds1 - mu2 * dS_ds12;
        const size_t o = c * plane + (size_t)gy * W + gx, cs = (size_t)C * plane;   // dmaps: (3 maps, C, H, W)
        dmaps[o] = dS_dmu1;
        dmaps[cs + o] = dS_ds1;
        dmaps[2 * cs + o] = dS_ds12;
        ss += S;
        l1 += fabsf(xs[j] - ys[j]);
      }
    }
    __syncthreads();   // the next tile overwrites hm
  }
  // block sums -> one (l1, ssim) pair per block, summed later in a fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { l1 += __shfl_xor(l1, o); ss += __shfl_xor(ss, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = l1; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// out = (loss, l1 mean, ssim mean); one block, fixed summation order, double accumulation.
__global__ void __launch_bounds__(1024) loss_finalize_kernel(int nblocks, const float* __restrict__ partial, double n,
                                                             float lambda, float* __restrict__ out) {
  __shared__ double r0[16], r1[16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 1024) {
    const float2 p = reinterpret_cast<const float2*>(partial)[i];
    a += (double)p.x; b += (double)p.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = a; r1[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < 16; k++) { s0 += r0[k]; s1 += r1[k]; }
    const double l1 = s0 / n, ss = s1 / n;
    out[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ss));
    out[1] = (float)l1;
    out[2] = (float)ss;
  }
}

__global__ void __launch_bounds__(256, GSR_LOSS_BWD_PER_CU) loss_bwd_kernel(int H, int W, int C, const float* __restrict__ img,
                                                       const float* __restrict__ gt, const float* __restrict__ dmaps,
                                                       Window win, const float* __restrict__ grad_out, float lambda,
                                                       float inv_n, float* __restrict__ grad_img) {
  __shared__ __align__(16) float sm[3][kIH * kIS];
  __shared__ __align__(16) float hm[3][kIH * kTX];
  const size_t plane = (size_t)H * W, cs = (size_t)C * plane;
  const TileWalk walk = make_walk(H, W, C);
  const int lx = threadIdx.x % kTX, ly = (threadIdx.x / kTX) * kOut;
  const float go = grad_out[0];
  Halo pm[3];
  int c, bx, by;
  if (walk.first < walk.last) {
    walk.decode(walk.first, c, bx, by);
#pragma unroll
    for (int m = 0; m < 3; m++) fetch_halo(dmaps + m * cs + c * plane, H, W, bx * kTX - kR, by * kTY - kR, pm[m]);
  }
  for (int t = walk.first; t < walk.last; t += walk.step) {
    walk.decode(t, c, bx, by);
#pragma unroll
    for (int m = 0; m < 3; m++) stash_halo(pm[m], sm[m]);
    __syncthreads();
    if (t + walk.step < walk.last) {
      int nc, nbx, nby;
      walk.decode(t + walk.step, nc, nbx, nby);
#pragma unroll
      for (int m = 0; m < 3; m++) fetch_halo(dmaps + m * cs + nc * plane, H, W, nbx * kTX - kR, nby * kTY - kR, pm[m]);
    }
    // this tile's pixels of both images, needed only at the very end
    const int gx = bx * kTX + lx;
    float xs[kOut], ys[kOut];
#pragma unroll
    for (int j = 0; j < kOut; j++) {
      const int gy = by * kTY + ly + j;
      const bool in = gx < W && gy < H;
      const size_t o = c * plane + (size_t)gy * W + gx;
      xs[j] = in ? img[o] : 0.f;
      ys[j] = in ? gt[o] : 0.f;
    }
    __builtin_amdgcn_s_setprio(2);      // as in the forward
    for (int k = threadIdx.x; k < kHTasks; k += 256) {
      const int r = k / (kTX / kOut), q = (k % (kTX / kOut)) * kOut;
#pragma unroll
      for (int m = 0; m < 3; m++) {
        float a[kNeed], o[kOut];
        read14(sm[m] + r * kIS + q, a);
#pragma unroll
        for (int j = 0; j < kOut; j++) {
          float s = 0.f;
#pragma unroll
          for (int kk = 0; kk < kTaps; kk++) s += win.w[kk] * a[j + kk];
          o[j] = s;
        }
        *reinterpret_cast<float4*>(&hm[m][r * kTX + q]) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    float acc[3][kOut];
#pragma unroll
    for (int m = 0; m < 3; m++) {
      float v[kNeed];
#pragma unroll
      for (int i = 0; i < kNeed; i++) v[i] = hm[m][(ly + i) * kTX + lx];
#pragma unroll
      for (int j = 0; j < kOut; j++) {
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < kTaps; kk++) s += win.w[kk] * v[j + kk];
        acc[m][j] = s;
      }
    }
#pragma unroll
    for (int j = 0; j < kOut; j++) {
      const int gy = by * kTY + ly + j;
      if (gx < W && gy < H) {
        const size_t o = c * plane + (size_t)gy * W + gx;
        const float x = xs[j], y = ys[j];
        const float d = x - y;
        const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        const float dssim = acc[0][j] + 2.0f * x * acc[1][j] + y * acc[2][j];     // d(sum of SSIM map) / dx
        grad_img[o] = go * inv_n * ((1.0f - lambda) * sgn - lambda * dssim);
      }
    }
    __syncthreads();   // the next tile overwrites sm, hm
  }
}

// Blocks launched for an image: what the 256 CUs hold at once (per_cu blocks each), a multiple of 8.
int loss_blocks(int C, int H, int W, int per_cu) {
  const int tiles = C * gsr::div_up(H, kTY) * gsr::div_up(W, kTX);
  if (tiles < 8) return tiles;
  const int cap = 256 * per_cu;
  return tiles < cap ? tiles / 8 * 8 : cap;
}
constexpr int kFwdPerCu = GSR_LOSS_FWD_PER_CU, kBwdPerCu = GSR_LOSS_BWD_PER_CU;   // resident blocks per CU (registers forward, LDS backward)

}  // namespace

extern "C" size_t gsr_loss_partial_count(int C, int H, int W) { return (size_t)loss_blocks(C, H, W, kFwdPerCu); }

extern "C" int gsr_loss_fwd(int C, int H, int W, const float* img, const float* gt, float lambda_dssim, float* dmaps,
                            float* partial, float* out3, void* stream) {
  GSR_REQUIRE(C == 3 && H > 0 && W > 0, "loss_fwd: expects (3,H,W), got C=%d H=%d W=%d", C, H, W);
  GSR_REQUIRE(img && gt && dmaps && partial && out3, "loss_fwd: null pointer");
  static const Window win = make_window();
  hipStream_t st = (hipStream_t)stream;
  const int blocks = loss_blocks(C, H, W, kFwdPerCu);
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(blocks), dim3(256), 0, st, H, W, C, img, gt, win, dmaps, partial);
  int rc = gsr::check_launch("loss_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, blocks, partial, (double)C * H * W,
                     lambda_dssim, out3);
  return gsr::check_launch("loss_fwd: finalize");
}

extern "C" int gsr_loss_bwd(int C, int H, int W, const float* img, const float* gt, const float* dmaps,
                            const float* grad_out, float lambda_dssim, float* grad_img, void* stream) {
  GSR_REQUIRE(C == 3 && H > 0 && W > 0, "loss_bwd: expects (3,H,W), got C=%d H=%d W=%d", C, H, W);
  GSR_REQUIRE(img && gt && dmaps && grad_out && grad_img, "loss_bwd: null pointer");
  static const Window win = make_window();
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(loss_blocks(C, H, W, kBwdPerCu)), dim3(256), 0, (hipStream_t)stream, H, W, C, img, gt,
                     dmaps, win, grad_out, lambda_dssim, 1.0f / ((float)C * (float)H * (float)W), grad_img);
  return gsr::check_launch("loss_bwd");
}
