// Per-Gaussian stages of the rasterizer for gfx950: forward preprocess
// (cull, project, covariance, EWA, conic, radius, tile rectangle, SH -> RGB)
// and its backward.  One lane per Gaussian; both kernels are HBM-bound
// (236 B in / ~80 B out per Gaussian forward at SH degree 3), so the work is
// organised around wide loads and full-line stores, not arithmetic.
//
// This translation unit is compiled with -ffp-contract=off: every + - * / and
// sqrt below is one IEEE-754 fp32 operation in the order written, so the
// forward results (and with them radii, tile rectangles and sort keys) are
// reproducible bit for bit by any IEEE implementation running the same
// sequence — which is how the CPU oracle pins them.
#include "gsr_common.h"

namespace {

constexpr int kBlock = 256;

struct Vec3 { float x, y, z; };

// [x y z 1] * M for a row-major (4,4) M: the reference stores its camera
// matrices transposed and multiplies row vectors
// (scene/cameras.py:54-56, utils/graphics_utils.py:22-29).
__device__ __forceinline__ Vec3 xform43(const float* __restrict__ m, float x, float y, float z) {
  Vec3 o;
  o.x = m[0] * x + m[4] * y + m[8] * z + m[12];
  o.y = m[1] * x + m[5] * y + m[9] * z + m[13];
  o.z = m[2] * x + m[6] * y + m[10] * z + m[14];
  return o;
}

// Rotation matrix of quaternion (w,x,y,z), taken as given (the reference
// normalises in get_rotation before the call, scene/gaussian_model.py:100-101).
__device__ __forceinline__ void quat_to_rot(float r, float x, float y, float z, float* R) {
  R[0] = 1.0f - 2.0f * (y * y + z * z);
  R[1] = 2.0f * (x * y - r * z);
  R[2] = 2.0f * (x * z + r * y);
  R[3] = 2.0f * (x * y + r * z);
  R[4] = 1.0f - 2.0f * (x * x + z * z);
  R[5] = 2.0f * (y * z - r * x);
  R[6] = 2.0f * (x * z - r * y);
  R[7] = 2.0f * (y * z + r * x);
  R[8] = 1.0f - 2.0f * (x * x + y * y);
}

// Sigma = L L^T with L = R diag(mod * s), packed (xx,xy,xz,yy,yz,zz)
// (Python twin: scene/gaussian_model.py:27-31, utils/general_utils.py:64-110).
__device__ __forceinline__ void cov3d_from_scale_rot(const float* s, float mod, const float* q,
                                                     float* c6) {
  float R[9], L[9];
  quat_to_rot(q[0], q[1], q[2], q[3], R);
  const float sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    L[3 * i + 0] = R[3 * i + 0] * sx;
    L[3 * i + 1] = R[3 * i + 1] * sy;
    L[3 * i + 2] = R[3 * i + 2] * sz;
  }
  c6[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
  c6[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
  c6[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
  c6[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
  c6[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
  c6[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
}

struct Ewa {
  float tx, ty, tz, xmul, ymul;
  float M0[3], M1[3];
  float a, b, c;
};

// EWA splatting: 2D covariance J W Sigma W^T J^T (+ dilation) of a Gaussian
// whose view-space mean is pv.
__device__ __forceinline__ void ewa_project(const float* __restrict__ V, int W, int H, float tanfovx,
                                            float tanfovy, Vec3 pv, const float* c6, Ewa& e) {
  const float fx = (float)W / (2.0f * tanfovx);
  const float fy = (float)H / (2.0f * tanfovy);
  const float limx = GSR_FRUSTUM_CLAMP * tanfovx;
  const float limy = GSR_FRUSTUM_CLAMP * tanfovy;
  const float tz = pv.z;
  const float txtz = pv.x / tz, tytz = pv.y / tz;
  e.xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
  e.ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
  const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
  const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
  e.tx = tx; e.ty = ty; e.tz = tz;
  const float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
  const float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const float W0 = V[4 * j + 0], W1 = V[4 * j + 1], W2 = V[4 * j + 2];
    e.M0[j] = J00 * W0 + J02 * W2;
    e.M1[j] = J11 * W1 + J12 * W2;
  }
  const float* M0 = e.M0;
  const float* M1 = e.M1;
  float v0[3], v1[3];
  v0[0] = c6[0] * M0[0] + c6[1] * M0[1] + c6[2] * M0[2];
  v0[1] = c6[1] * M0[0] + c6[3] * M0[1] + c6[4] * M0[2];
  v0[2] = c6[2] * M0[0] + c6[4] * M0[1] + c6[5] * M0[2];
  v1[0] = c6[0] * M1[0] + c6[1] * M1[1] + c6[2] * M1[2];
  v1[1] = c6[1] * M1[0] + c6[3] * M1[1] + c6[4] * M1[2];
  v1[2] = c6[2] * M1[0] + c6[4] * M1[1] + c6[5] * M1[2];
  e.a = (M0[0] * v0[0] + M0[1] * v0[1] + M0[2] * v0[2]) + GSR_COV2D_DILATION;
  e.b = M0[0] * v1[0] + M0[1] * v1[1] + M0[2] * v1[2];
  e.c = (M1[0] * v1[0] + M1[1] * v1[1] + M1[2] * v1[2]) + GSR_COV2D_DILATION;
}

// Real SH basis of a unit direction; b[k] multiplies coefficient k
// (Python twin: utils/sh_utils.py:74-100).
template <int D>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* b) {
  b[0] = GSR_SH_C0;
  if (D > 0) {
    b[1] = -GSR_SH_C1 * y;
    b[2] = GSR_SH_C1 * z;
    b[3] = -GSR_SH_C1 * x;
  }
  if (D > 1) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = GSR_SH_C2_0 * xy;
    b[5] = GSR_SH_C2_1 * yz;
    b[6] = GSR_SH_C2_2 * (2.0f * zz - xx - yy);
    b[7] = GSR_SH_C2_3 * xz;
    b[8] = GSR_SH_C2_4 * (xx - yy);
    if (D > 2) {
      b[9] = GSR_SH_C3_0 * y * (3.0f * xx - yy);
      b[10] = GSR_SH_C3_1 * xy * z;
      b[11] = GSR_SH_C3_2 * y * (4.0f * zz - xx - yy);
      b[12] = GSR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
      b[13] = GSR_SH_C3_4 * x * (4.0f * zz - xx - yy);
      b[14] = GSR_SH_C3_5 * z * (xx - yy);
      b[15] = GSR_SH_C3_6 * x * (xx - 3.0f * yy);
    }
  }
}

// Loads the first N = (D+1)^2 coefficients x 3 channels of Gaussian i into
// sh[3*k + c].  With the reference's (P,16,3) layout each Gaussian is 192 B,
// 16-byte aligned: 16-byte loads.  Other M: 4-byte loads.
// When shs_rest is given the coefficients come from the model's two parameter
// tensors instead: shs = _features_dc (P,1,3), shs_rest = _features_rest
// (P,n_rest,3) (scene/gaussian_model.py:108-111 concatenates them per call).
// rest_row points at this Gaussian's row of _features_rest: n_rest*12 bytes, 4-byte aligned
// only, so it is read with 16-byte loads declared as 4-byte aligned (gfx950 global loads
// need dword alignment, not natural alignment).
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
template <int D>
__device__ __forceinline__ void load_sh(const float* __restrict__ shs, const float* __restrict__ rest_row, int M,
                                        int i, float* sh) {
  constexpr int N = (D + 1) * (D + 1);
  if (rest_row) {
    const float* dc = shs + (size_t)i * 3;
    sh[0] = dc[0]; sh[1] = dc[1]; sh[2] = dc[2];
    constexpr int R = 3 * N - 3, Q = R / 4;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const float4u v = *reinterpret_cast<const float4u*>(rest_row + 4 * k);
      sh[3 + 4 * k] = v.x; sh[4 + 4 * k] = v.y; sh[5 + 4 * k] = v.z; sh[6 + 4 * k] = v.w;
    }
#pragma unroll
    for (int k = 4 * Q; k < R; k++) sh[3 + k] = rest_row[k];
  } else if (M == 16) {
    const float4* p = reinterpret_cast<const float4*>(shs) + (size_t)i * 12;
    constexpr int Q = (3 * N + 3) / 4;
    float4 v[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) v[k] = p[k];
#pragma unroll
    for (int k = 0; k < 3 * N; k++) {
      const float4 q = v[k / 4];
      sh[k] = (k % 4 == 0) ? q.x : (k % 4 == 1) ? q.y : (k % 4 == 2) ? q.z : q.w;
    }
  } else {
    const float* p = shs + (size_t)i * M * 3;
#pragma unroll
    for (int k = 0; k < 3 * N; k++) sh[k] = p[k];
  }
}

// Raw mode: a block's kBlock rows of a (P, n_rest, 3) tensor are contiguous in memory
// (n_rest*12 bytes each, 4-byte aligned only).  Each lane fills row threadIdx.x of an LDS
// tile, which then goes to HBM with fully coalesced dword stores.
__device__ __forceinline__ void tile_store_rows(float* __restrict__ dst, int P, int R, const float* tile) {
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kBlock * R;
  const int rows = min(kBlock, P - (int)blockIdx.x * kBlock);
  for (int e = threadIdx.x; e < rows * R; e += kBlock) dst[base + e] = tile[e];
}

// The same for rows of an even number of words, which the lanes would write into the same few LDS banks:
// the tile's rows are R + 1 words apart, and the copy undoes the padding.
template <int R>
__device__ __forceinline__ void tile_store_padded_rows(float* __restrict__ dst, int P, const float* tile) {
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * kBlock * R;
  const int rows = min(kBlock, P - (int)blockIdx.x * kBlock);
  for (int e = threadIdx.x; e < rows * R; e += kBlock) dst[base + e] = tile[(e / R) * (R + 1) + e % R];
}

// Activations of the model's raw parameters (scene/gaussian_model.py:33-41,95-118):
// scale = exp(raw), opacity = sigmoid(raw), rotation = raw / max(|raw|, 1e-12).
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float quat_norm(const float* q) {
  return fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
}

struct FwdArgs {
  uint32_t* clear_at; size_t clear_words;   // memory of the caller's that the workgroups clear as their last act (may be none)
  int raw, n_rest, reference_lists, P, M, W, H;
  const float* means3D; const float* scales; const float* rotations; const float* opacities;
  const float* shs; const float* shs_rest; const float* colors_precomp;
  const float* cov3D_precomp; const float* view; const float* proj; const float* campos;
  float scale_modifier, tanfovx, tanfovy;
  gsr::Pack pack;
  uint32_t* records; int32_t* radii; uint32_t* tiles_touched; uint32_t* depth_keys; uint32_t* rects;
  unsigned long long* sort_vals; uint32_t* partials; uint8_t* visible;
};

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return gsr::wave_sum(v); }

// A lane's 64-byte record leaves the kernel through LDS: written as four 16-byte stores per lane the
// records of a wave would go out as 64 partial writes of 64-byte segments per instruction (one per lane,
// 16 bytes each), which costs more than the bytes (0.076 -> 0.061 ms for 1M Gaussians); transposed in
// LDS, every store instruction of a wave writes 1 KB of consecutive memory = whole records.
// Row pitch 5 x 16 bytes: the 16-byte writes of eight consecutive lanes then fall into different banks.
constexpr int kRecPitch = 5;

template <int D>
__global__ void __launch_bounds__(kBlock) preprocess_fwd_kernel(FwdArgs a) {
  __shared__ uint32_t s_part[kBlock / 64][4];
  __shared__ uint4 s_rec[kBlock / 64][64 * kRecPitch];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  uint4 rec0 = make_uint4(0u, 0u, 0u, 0u), rec1 = rec0, rec2 = rec0, rec3 = rec0;   // culled: all zero (radius 0)
  bool vis = false, binned = false;
  int radius = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;   // the rectangle the tile lists are built from
  float pxs = 0.f, pys = 0.f, conA = 0.f, conB = 0.f, conC = 0.f, hx = -1.f, hy = -1.f;
  float opacity = 0.f, depth = 0.f;
  if (i < a.P) {
    const float px3 = a.means3D[3 * i], py3 = a.means3D[3 * i + 1], pz3 = a.means3D[3 * i + 2];
    const Vec3 pv = xform43(a.view, px3, py3, pz3);
    depth = pv.z;
    vis = pv.z > GSR_NEAR_CULL;
    if (vis) {
      const float* PV = a.proj;
      const float hxh = PV[0] * px3 + PV[4] * py3 + PV[8] * pz3 + PV[12];
      const float hyh = PV[1] * px3 + PV[5] * py3 + PV[9] * pz3 + PV[13];
      const float hwh = PV[3] * px3 + PV[7] * py3 + PV[11] * pz3 + PV[15];
      const float pw = 1.0f / (hwh + GSR_W_EPS);
      const float ndcx = hxh * pw, ndcy = hyh * pw;

      float c6[6];
      if (a.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * (size_t)i + k];
      } else {
        float s[3] = {a.scales[3 * i], a.scales[3 * i + 1], a.scales[3 * i + 2]};
        const float4 qv = reinterpret_cast<const float4*>(a.rotations)[i];
        float q[4] = {qv.x, qv.y, qv.z, qv.w};
        if (a.raw) {
          s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]);
          const float n = quat_norm(q);
          q[0] = q[0] / n; q[1] = q[1] / n; q[2] = q[2] / n; q[3] = q[3] / n;
        }
        cov3d_from_scale_rot(s, a.scale_modifier, q, c6);
      }
      Ewa e;
      ewa_project(a.view, a.W, a.H, a.tanfovx, a.tanfovy, pv, c6, e);
      const float det = e.a * e.c - e.b * e.b;
      vis = det != 0.0f;
      if (vis) {
        const float det_inv = 1.0f / det;
        conA = e.c * det_inv; conB = -e.b * det_inv; conC = e.a * det_inv;
        const float mid = 0.5f * (e.a + e.c);
        const float disc = sqrtf(fmaxf(GSR_LAMBDA_FLOOR, mid * mid - det));
        const float lam1 = mid + disc, lam2 = mid - disc;
        const float rf = ceilf(GSR_RADIUS_SIGMAS * sqrtf(fmaxf(lam1, lam2)));
        radius = (rf < 2147483520.0f) ? (int)rf : 2147483647;
        pxs = ((ndcx + 1.0f) * (float)a.W - 1.0f) * 0.5f;
        pys = ((ndcy + 1.0f) * (float)a.H - 1.0f) * 0.5f;
        const int gx = (a.W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (a.H + GSR_TILE_Y - 1) / GSR_TILE_Y;
        const float rr = (float)radius;
        x0 = (int)fminf((float)gx, fmaxf(0.0f, (pxs - rr) / (float)GSR_TILE_X));
        y0 = (int)fminf((float)gy, fmaxf(0.0f, (pys - rr) / (float)GSR_TILE_Y));
        x1 = (int)fminf((float)gx, fmaxf(0.0f, (pxs + rr + (float)(GSR_TILE_X - 1)) / (float)GSR_TILE_X));
        y1 = (int)fminf((float)gy, fmaxf(0.0f, (pys + rr + (float)(GSR_TILE_Y - 1)) / (float)GSR_TILE_Y));
        vis = (x1 - x0) * (y1 - y0) > 0;
        opacity = a.raw ? sigmoidf(a.opacities[i]) : a.opacities[i];
        // Half extents of the region where alpha can reach 1/255: the ellipse
        // d^T conic d <= 2 ln(255 o) spans sqrt(2 ln(255 o) * cov_xx) in x.
        // Widened a little so that rounding can never make it exclusive.
        const float tau = 2.0f * __logf(255.0f * opacity);
        if (tau > 0.0f) {
          hx = sqrtf(tau * e.a) * 1.0005f + 0.01f;
          hy = sqrtf(tau * e.c) * 1.0005f + 0.01f;
        }
        // The module bins every tile of the 3-sigma rectangle.  A pixel can blend the splat only
        // inside that ellipse, so tiles of the rectangle that its bounding box misses hold no
        // contribution: unless the module's own lists are asked for they are left out (images,
        // radii and gradients do not change; the lists get shorter, a third at typical opacities).
        bx0 = x0; by0 = y0; bx1 = x1; by1 = y1;
        if (!a.reference_lists) {
          if (hx < 0.0f) {
            bx1 = bx0;
          } else {
            bx0 = max(bx0, (int)fminf((float)gx, fmaxf(0.0f, floorf((pxs - hx) / (float)GSR_TILE_X))));
            by0 = max(by0, (int)fminf((float)gy, fmaxf(0.0f, floorf((pys - hy) / (float)GSR_TILE_Y))));
            bx1 = min(bx1, (int)fminf((float)gx, fmaxf(0.0f, floorf((pxs + hx) / (float)GSR_TILE_X) + 1.0f)));
            by1 = min(by1, (int)fminf((float)gy, fmaxf(0.0f, floorf((pys + hy) / (float)GSR_TILE_Y) + 1.0f)));
          }
        }
        binned = vis && bx1 > bx0 && by1 > by0;
      }
    }
    if (a.visible) a.visible[i] = vis ? 1 : 0;
    if (!vis) {
      a.radii[i] = 0;
      if (a.tiles_touched) a.tiles_touched[i] = 0;
      a.depth_keys[i] = 0xFFFFFFFFu;
      if (a.rects) reinterpret_cast<uint2*>(a.rects)[i] = make_uint2(0u, 0u);
      a.sort_vals[i] = (unsigned long long)(uint32_t)i;
    } else {
      const uint32_t rmin = (uint32_t)x0 | ((uint32_t)y0 << 16), rmax = (uint32_t)x1 | ((uint32_t)y1 << 16);
      a.radii[i] = radius;
      if (a.tiles_touched) a.tiles_touched[i] = (uint32_t)((x1 - x0) * (y1 - y0));
      a.depth_keys[i] = binned ? __float_as_uint(depth) : 0xFFFFFFFFu;
      if (a.rects) reinterpret_cast<uint2*>(a.rects)[i] = make_uint2(rmin, rmax);
      a.sort_vals[i] = binned ? a.pack.pack((uint32_t)i, (uint32_t)bx0, (uint32_t)bx1 - 1u, (uint32_t)by0, (uint32_t)by1 - 1u)
                              : (unsigned long long)(uint32_t)i;
      rec0 = make_uint4(__float_as_uint(pxs), __float_as_uint(pys), __float_as_uint(hx), __float_as_uint(hy));
      rec1 = make_uint4(__float_as_uint(conA), __float_as_uint(conB), __float_as_uint(conC), __float_as_uint(opacity));
      rec3 = make_uint4(__float_as_uint(depth), rmin, rmax, (uint32_t)radius);

      // colour: SH -> RGB with +0.5 and clamp at 0, or the colour handed in
      float r = 0.f, g = 0.f, b = 0.f;
      unsigned clamped = 0;
      if (a.colors_precomp) {
        r = a.colors_precomp[3 * (size_t)i];
        g = a.colors_precomp[3 * (size_t)i + 1];
        b = a.colors_precomp[3 * (size_t)i + 2];
      } else {
        // gaussian_renderer/__init__.py:75-78
        float dx = px3 - a.campos[0], dy = py3 - a.campos[1], dz = pz3 - a.campos[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        dx = dx / len; dy = dy / len; dz = dz / len;
        constexpr int N = (D + 1) * (D + 1);
        float bas[N], sh[3 * N], col[3];
        sh_basis<D>(dx, dy, dz, bas);
        load_sh<D>(a.shs, a.shs_rest ? a.shs_rest + (size_t)i * (a.n_rest * 3) : nullptr, a.M, i, sh);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          float acc = bas[0] * sh[ch];
#pragma unroll
          for (int k = 1; k < N; k++) acc = acc + bas[k] * sh[3 * k + ch];
          acc = acc + GSR_SH_OFFSET;
          if (acc < 0.0f) clamped |= 1u << ch;
          col[ch] = fmaxf(acc, 0.0f);
        }
        r = col[0]; g = col[1]; b = col[2];
      }
      rec2 = make_uint4(__float_as_uint(r), __float_as_uint(g), __float_as_uint(b), clamped);
    }
  }
  {
    // the wave's 64 records: every lane deposits its own, then store j of the wave writes records 16j .. 16j+15,
    // lane l the quarter l % 4 of record 16j + l / 4 (a wave's LDS operations execute in order)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* mine = &s_rec[wave][lane * kRecPitch];
    mine[0] = rec0; mine[1] = rec1; mine[2] = rec2; mine[3] = rec3;
    const int first = i - lane;                                    // the wave's first Gaussian
    uint4* out = reinterpret_cast<uint4*>(a.records) + (size_t)first * 4 + lane;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int r = 16 * j + (lane >> 2);
      const uint4 v = s_rec[wave][r * kRecPitch + (lane & 3)];
      // (nontemporal: the records are next read by the blend kernels, three launches on, by gathers that find little
      // of 64 MB in the L2 anyway; as ordinary stores they cost this kernel 7 us of 72 at 1M Gaussians.  The same hint
      // on the kernel's loads makes it half as slow again: a lane's 192-byte row arrives 16 bytes at a time, and the
      // other pieces of a line must wait for it in the cache.)
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      const u4 vv = {v.x, v.y, v.z, v.w};
      if (first + r < a.P) __builtin_nontemporal_store(vv, reinterpret_cast<u4*>(out + 64 * j));
    }
  }
  // list sizes of the binning (tile instances, supercell entries, cell entries, visible Gaussians):
  // one partial record per block, summed by a second small kernel; no atomics
  uint32_t n = 0, e1 = 0, e2 = 0, v = 0;
  if (binned) {
    const uint32_t strips = (uint32_t)((bx1 - 1) / GSR_SUPERCELL - bx0 / GSR_SUPERCELL + 1);
    const uint32_t bands = (uint32_t)((by1 - 1) / GSR_SUPERCELL - by0 / GSR_SUPERCELL + 1);
    n = (uint32_t)((bx1 - bx0) * (by1 - by0));
    e1 = strips * bands;
    e2 = strips * (uint32_t)(by1 - by0);
    v = 1u;
  }
  n = wave_sum_u32(n); e1 = wave_sum_u32(e1); e2 = wave_sum_u32(e2); v = wave_sum_u32(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { s_part[wave][0] = n; s_part[wave][1] = e1; s_part[wave][2] = e2; s_part[wave][3] = v; }
  __syncthreads();
  if (threadIdx.x < 4) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) t += s_part[w][threadIdx.x];
    a.partials[4 * blockIdx.x + threadIdx.x] = t;
  }
  for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < a.clear_words; k += (size_t)gridDim.x * kBlock) a.clear_at[k] = 0u;
}

__global__ void __launch_bounds__(kBlock) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                              const float* __restrict__ view,
                                                              uint8_t* __restrict__ present) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= P) return;
  const Vec3 pv = xform43(view, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
  present[i] = pv.z > GSR_NEAR_CULL;
}

struct BwdArgs {
  int sparse;   // the outputs are zero already: only Gaussians with gradients write
  const uint8_t* touched;   // sparse mode, optional: 0 = the Gaussian's row of sums is all zeros (it is not read then)
  int raw, n_rest; const float* shs_rest; const float* opacity_raw; float* dL_dshs_rest;
  int P, M, W, H;
  const float* means3D; const float* scales; const float* rotations; const float* shs;
  const float* cov3D_precomp; const float* view; const float* proj; const float* campos;
  float scale_modifier, tanfovx, tanfovy;
  const uint32_t* records; const float* dL_dsplat;
  float* dL_dmeans3D; float* dL_dmeans2D; float* dL_dopacities; float* dL_dshs; float* dL_dcolors;
  float* dL_dscales; float* dL_drotations; float* dL_dcov3D;
};

// Stores n consecutive floats (n a multiple of 4, 16-byte aligned) as float4.
template <int N>
__device__ __forceinline__ void store_f4(float* dst, const float* v) {
  float4* d = reinterpret_cast<float4*>(dst);
#pragma unroll
  for (int k = 0; k < N / 4; k++) d[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}

// The gradients of Gaussian i (valid: i < P).  SPARSE: the outputs were zero when the kernel started and only a
// Gaussian with gradients writes, each its own rows; otherwise every row is written, the SH gradient through the
// workgroup's LDS tile (rest_tile), which every thread of the workgroup must then reach.
template <int D, bool SPARSE>
__device__ __forceinline__ void bwd_one_gaussian(const BwdArgs& a, const int i, const bool valid, float* rest_tile) {
  // Every gradient of this kernel is linear in the nine sums the blend backward left for the Gaussian.  A
  // Gaussian that no pixel blended (culled, or hidden behind saturated pixels: most of a dense scene) has nine
  // zeros, hence zero gradients: nothing else of it is loaded, not even its record.  (NaN != 0, so a poisoned
  // row still takes the full path.)  A non-zero row of a Gaussian whose record says "culled" (only a caller
  // that fills the rows itself can produce one) counts as zero too.
  const float4* gsp = reinterpret_cast<const float4*>(a.dL_dsplat) + (size_t)i * (GSR_GRAD_WORDS / 4);
  float4 g0 = make_float4(0, 0, 0, 0), g1 = g0, g2 = g0;
  float4 m0 = g0, m1 = g0;
  bool vis = false;
  if (valid) {
    m0 = gsp[0]; m1 = gsp[1]; g2 = gsp[2];
    vis = (m0.x != 0.f) | (m0.y != 0.f) | (m0.z != 0.f) | (m0.w != 0.f) | (m1.x != 0.f) | (m1.y != 0.f) |
          (m1.z != 0.f) | (m1.w != 0.f) | (g2.x != 0.f);
    if (vis) vis = (int)a.records[(size_t)i * GSR_REC_WORDS + GSR_REC_RADIUS] > 0;
    if (!vis) g2 = make_float4(0, 0, 0, 0);
  }
  if (vis) {
    // The blend backward leaves moments of u = dL/dG * G over each splat's pixels:
    //   (sum u, sum u dx, sum u dy, sum u dx^2, sum u dx dy, sum u dy^2, dr, dg, db)
    // with d = splat centre - pixel.  With the conic (A,B,C) and opacity o of the record:
    //   dL/dopacity = sum u / o
    //   dL/dmean2D  = -(A sum u dx + B sum u dy) * W/2,  -(C sum u dy + B sum u dx) * H/2
    //   dL/d(A,B,C) = (-1/2 sum u dx^2, -sum u dx dy, -1/2 sum u dy^2)
    const uint4 r1 = reinterpret_cast<const uint4*>(a.records)[(size_t)i * 4 + 1];
    const float cA = __uint_as_float(r1.x), cB = __uint_as_float(r1.y), cC = __uint_as_float(r1.z);
    const float op = __uint_as_float(r1.w);
    g0.x = -(cA * m0.y + cB * m0.z) * (0.5f * (float)a.W);
    g0.y = -(cC * m0.z + cB * m0.y) * (0.5f * (float)a.H);
    g0.z = -0.5f * m0.w;
    g0.w = -m1.x;
    g1.x = -0.5f * m1.y;
    g1.y = op > 0.0f ? m0.x / op : 0.0f;   // opacity 0 blends nowhere: sum u is 0 and so is the gradient, not 0/0
    g1.z = m1.z;
    g1.w = m1.w;
  }
  // g0 = (dmean2D.x, dmean2D.y, dA, dB); g1 = (dC, dopacity, dr, dg); g2 = (db, -, -, -)
  float dmean[3] = {0.f, 0.f, 0.f};
  float dc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int N = (D + 1) * (D + 1);
  float dsh[48];
#pragma unroll
  for (int k = 0; k < 48; k++) dsh[k] = 0.f;

  if (vis) {
    const float px3 = a.means3D[3 * i], py3 = a.means3D[3 * i + 1], pz3 = a.means3D[3 * i + 2];
    const float* V = a.view;
    const float* PV = a.proj;
    const Vec3 pv = xform43(V, px3, py3, pz3);
    float c6[6];
    float s[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, qn = 1.0f;
    if (a.cov3D_precomp) {
#pragma unroll
      for (int k = 0; k < 6; k++) c6[k] = a.cov3D_precomp[6 * (size_t)i + k];
    } else {
      s[0] = a.scales[3 * i]; s[1] = a.scales[3 * i + 1]; s[2] = a.scales[3 * i + 2];
      const float4 qv = reinterpret_cast<const float4*>(a.rotations)[i];
      q[0] = qv.x; q[1] = qv.y; q[2] = qv.z; q[3] = qv.w;
      if (a.raw) {
        s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]);
        qn = quat_norm(q);
        q[0] = q[0] / qn; q[1] = q[1] / qn; q[2] = q[2] / qn; q[3] = q[3] / qn;
      }
      cov3d_from_scale_rot(s, a.scale_modifier, q, c6);
    }
    Ewa e;
    ewa_project(V, a.W, a.H, a.tanfovx, a.tanfovy, pv, c6, e);
    const float fx = (float)a.W / (2.0f * a.tanfovx);
    const float fy = (float)a.H / (2.0f * a.tanfovy);

    // (a) conic -> 2D covariance -> 3D covariance and view-space mean
    const float ca = e.a, cb = e.b, cc = e.c;
    const float dA = g0.z, dB = g0.w, dC = g1.x;
    const float det = ca * cc - cb * cb;
    const float det2inv = 1.0f / (det * det + GSR_DET2_EPS);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    if (det2inv != 0.0f) {
      dL_da = det2inv * (-cc * cc * dA + cb * cc * dB + (det - ca * cc) * dC);
      dL_dc = det2inv * (-ca * ca * dC + ca * cb * dB + (det - ca * cc) * dA);
      dL_db = det2inv * (2.0f * cb * cc * dA - (det + 2.0f * cb * cb) * dB + 2.0f * ca * cb * dC);
    }
    const float* M0 = e.M0;
    const float* M1 = e.M1;
    dc6[0] = M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
    dc6[3] = M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
    dc6[5] = M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
    dc6[1] = 2.0f * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + 2.0f * M1[0] * M1[1] * dL_dc;
    dc6[2] = 2.0f * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + 2.0f * M1[0] * M1[2] * dL_dc;
    dc6[4] = 2.0f * M0[1] * M0[2] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + 2.0f * M1[1] * M1[2] * dL_dc;

    float v0[3], v1[3];
    v0[0] = c6[0] * M0[0] + c6[1] * M0[1] + c6[2] * M0[2];
    v0[1] = c6[1] * M0[0] + c6[3] * M0[1] + c6[4] * M0[2];
    v0[2] = c6[2] * M0[0] + c6[4] * M0[1] + c6[5] * M0[2];
    v1[0] = c6[0] * M1[0] + c6[1] * M1[1] + c6[2] * M1[2];
    v1[1] = c6[1] * M1[0] + c6[3] * M1[1] + c6[4] * M1[2];
    v1[2] = c6[2] * M1[0] + c6[4] * M1[1] + c6[5] * M1[2];
    float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float dM0 = 2.0f * v0[j] * dL_da + v1[j] * dL_db;
      const float dM1 = 2.0f * v1[j] * dL_dc + v0[j] * dL_db;
      dJ00 += V[4 * j + 0] * dM0; dJ02 += V[4 * j + 2] * dM0;
      dJ11 += V[4 * j + 1] * dM1; dJ12 += V[4 * j + 2] * dM1;
    }
    const float tz = 1.0f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = e.xmul * -fx * tz2 * dJ02;
    const float dty = e.ymul * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.0f * fx * e.tx) * tz3 * dJ02 +
                      (2.0f * fy * e.ty) * tz3 * dJ12;
#pragma unroll
    for (int j = 0; j < 3; j++) dmean[j] += V[4 * j + 0] * dtx + V[4 * j + 1] * dty + V[4 * j + 2] * dtz;

    // (b) screen-space mean -> 3D mean through the perspective divide
    const float hxh = PV[0] * px3 + PV[4] * py3 + PV[8] * pz3 + PV[12];
    const float hyh = PV[1] * px3 + PV[5] * py3 + PV[9] * pz3 + PV[13];
    const float hwh = PV[3] * px3 + PV[7] * py3 + PV[11] * pz3 + PV[15];
    const float mw = 1.0f / (hwh + GSR_W_EPS);
    const float mul1 = hxh * mw * mw, mul2 = hyh * mw * mw;
#pragma unroll
    for (int j = 0; j < 3; j++)
      dmean[j] += (PV[4 * j + 0] * mw - PV[4 * j + 3] * mul1) * g0.x + (PV[4 * j + 1] * mw - PV[4 * j + 3] * mul2) * g0.y;

    // (c) colour -> SH coefficients and view direction
    if (a.shs) {
      const float dirx = px3 - a.campos[0], diry = py3 - a.campos[1], dirz = pz3 - a.campos[2];
      const float len = sqrtf(dirx * dirx + diry * diry + dirz * dirz);
      const float x = dirx / len, y = diry / len, z = dirz / len;
      float bas[N], sh[3 * N];
      sh_basis<D>(x, y, z, bas);
      load_sh<D>(a.shs, a.shs_rest ? a.shs_rest + (size_t)i * (a.n_rest * 3) : nullptr, a.M, i, sh);
      const uint32_t clamped = a.records[(size_t)i * GSR_REC_WORDS + GSR_REC_CLAMPED];
      const float dRGB[3] = {(clamped & 1u) ? 0.f : g1.z, (clamped & 2u) ? 0.f : g1.w, (clamped & 4u) ? 0.f : g2.x};
#pragma unroll
      for (int k = 0; k < N; k++) {
        dsh[3 * k] = bas[k] * dRGB[0]; dsh[3 * k + 1] = bas[k] * dRGB[1]; dsh[3 * k + 2] = bas[k] * dRGB[2];
      }
      float ddx = 0.f, ddy = 0.f, ddz = 0.f;
      // s_k = sh_k . dRGB; d basis_k / d(x,y,z) written out per band
      auto sk = [&](int k) { return sh[3 * k] * dRGB[0] + sh[3 * k + 1] * dRGB[1] + sh[3 * k + 2] * dRGB[2]; };
      if (D > 0) {
        ddy += -GSR_SH_C1 * sk(1);
        ddz += GSR_SH_C1 * sk(2);
        ddx += -GSR_SH_C1 * sk(3);
      }
      if (D > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float s4 = sk(4), s5 = sk(5), s6 = sk(6), s7 = sk(7), s8 = sk(8);
        ddx += GSR_SH_C2_0 * y * s4 + GSR_SH_C2_2 * -2.0f * x * s6 + GSR_SH_C2_3 * z * s7 + GSR_SH_C2_4 * 2.0f * x * s8;
        ddy += GSR_SH_C2_0 * x * s4 + GSR_SH_C2_1 * z * s5 + GSR_SH_C2_2 * -2.0f * y * s6 + GSR_SH_C2_4 * -2.0f * y * s8;
        ddz += GSR_SH_C2_1 * y * s5 + GSR_SH_C2_2 * 4.0f * z * s6 + GSR_SH_C2_3 * x * s7;
        if (D > 2) {
          const float s9 = sk(9), s10 = sk(10), s11 = sk(11), s12 = sk(12), s13 = sk(13), s14 = sk(14), s15 = sk(15);
          ddx += GSR_SH_C3_0 * 6.0f * x * y * s9 + GSR_SH_C3_1 * y * z * s10 + GSR_SH_C3_2 * -2.0f * x * y * s11 +
                 GSR_SH_C3_3 * -6.0f * x * z * s12 + GSR_SH_C3_4 * (4.0f * zz - 3.0f * xx - yy) * s13 +
                 GSR_SH_C3_5 * 2.0f * x * z * s14 + GSR_SH_C3_6 * (3.0f * xx - 3.0f * yy) * s15;
          ddy += GSR_SH_C3_0 * (3.0f * xx - 3.0f * yy) * s9 + GSR_SH_C3_1 * x * z * s10 +
                 GSR_SH_C3_2 * (4.0f * zz - xx - 3.0f * yy) * s11 + GSR_SH_C3_3 * -6.0f * y * z * s12 +
                 GSR_SH_C3_4 * -2.0f * x * y * s13 + GSR_SH_C3_5 * -2.0f * y * z * s14 + GSR_SH_C3_6 * -6.0f * x * y * s15;
          ddz += GSR_SH_C3_1 * x * y * s10 + GSR_SH_C3_2 * 8.0f * y * z * s11 +
                 GSR_SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy) * s12 + GSR_SH_C3_4 * 8.0f * x * z * s13 +
                 GSR_SH_C3_5 * (xx - yy) * s14;
        }
      }
      const float nd = x * ddx + y * ddy + z * ddz;
      dmean[0] += (ddx - x * nd) / len;
      dmean[1] += (ddy - y * nd) / len;
      dmean[2] += (ddz - z * nd) / len;
    }

    // (d) 3D covariance -> scale, rotation
    if (!a.cov3D_precomp) {
      float R[9];
      quat_to_rot(q[0], q[1], q[2], q[3], R);
      const float mod = a.scale_modifier;
      const float S[3] = {mod * s[0], mod * s[1], mod * s[2]};
      const float Gm[9] = {dc6[0], 0.5f * dc6[1], 0.5f * dc6[2], 0.5f * dc6[1], dc6[3],
                           0.5f * dc6[4], 0.5f * dc6[2], 0.5f * dc6[4], dc6[5]};
      float dR[9];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        float ds = 0.f;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < 3; k++) acc += Gm[3 * r + k] * (R[3 * k + j] * S[j]);
          const float dLm = 2.0f * acc;
          ds += R[3 * r + j] * dLm;
          dR[3 * r + j] = S[j] * dLm;
        }
        dscale[j] = mod * ds;
      }
      const float r = q[0], x = q[1], y = q[2], z = q[3];
      drot[0] = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
      drot[1] = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]);
      drot[2] = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
      drot[3] = 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
      if (a.raw) {
        // back through exp and normalize: d/draw = dscale * scale; (g - q (q.g)) / |raw|
        dscale[0] *= s[0]; dscale[1] *= s[1]; dscale[2] *= s[2];
        const float qg = q[0] * drot[0] + q[1] * drot[1] + q[2] * drot[2] + q[3] * drot[3];
#pragma unroll
        for (int k = 0; k < 4; k++) drot[k] = (drot[k] - q[k] * qg) / qn;
      }
    }
  }
  float dopac = g1.y;
  if (a.raw && vis) {
    const float o = sigmoidf(a.opacity_raw[i]);
    dopac = dopac * (o * (1.0f - o));
  }

  if (SPARSE) {
    // the outputs were zero when the kernel started: only the Gaussians with gradients write, each its own rows
    if (!vis) return;
    // 16 bytes per store wherever the row allows it (dword alignment is all a global store needs): every store of a
    // lane goes to a line of its own here, and 48 stores of 4 bytes per lane were most of the kernel's requests
    if (a.dL_dshs_rest) {
      const int R = a.n_rest * 3;
      float* dst = a.dL_dshs_rest + (size_t)i * R;
#pragma unroll
      for (int k = 0; k < 44; k += 4) {
        if (k + 3 < R) {
          float4u v;
          v.x = dsh[3 + k]; v.y = dsh[4 + k]; v.z = dsh[5 + k]; v.w = dsh[6 + k];
          *reinterpret_cast<float4u*>(dst + k) = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (k + j < R) dst[k + j] = dsh[3 + k + j];
        }
      }
      if (44 < R) dst[44] = dsh[47];
      float* dc = a.dL_dshs + (size_t)i * 3;
      dc[0] = dsh[0]; dc[1] = dsh[1]; dc[2] = dsh[2];
    } else if (a.dL_dshs) {
      const int R = a.M * 3;
      float* dst = a.dL_dshs + (size_t)i * R;
#pragma unroll
      for (int k = 0; k < 48; k += 4) {
        if (k + 3 < R) {
          float4u v;
          v.x = dsh[k]; v.y = dsh[k + 1]; v.z = dsh[k + 2]; v.w = dsh[k + 3];
          *reinterpret_cast<float4u*>(dst + k) = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (k + j < R) dst[k + j] = dsh[k + j];
        }
      }
    }
  } else if (a.dL_dshs_rest) {
    // each lane puts its row into the LDS tile, then the block stores the tile fully coalesced
    const int R = a.n_rest * 3;
    float* row = rest_tile + threadIdx.x * R;
#pragma unroll
    for (int k = 3; k < 48; k++)
      if (k - 3 < R) row[k - 3] = dsh[k];
    tile_store_rows(a.dL_dshs_rest, a.P, R, rest_tile);
    if (valid) {
      float* dc = a.dL_dshs + (size_t)i * 3;
      dc[0] = dsh[0]; dc[1] = dsh[1]; dc[2] = dsh[2];
    }
  } else if (a.dL_dshs) {
    // the (P, M, 3) gradient the same way: written row by row from the lanes it would go out as 16-byte
    // pieces of 64 different rows per store instruction
    const int R = a.M * 3;
    if (R == 48) {
      float* row = rest_tile + threadIdx.x * 49;
#pragma unroll
      for (int k = 0; k < 48; k++) row[k] = dsh[k];
      tile_store_padded_rows<48>(a.dL_dshs, a.P, rest_tile);
    } else {
      float* row = rest_tile + threadIdx.x * R;
#pragma unroll
      for (int k = 0; k < 48; k++)
        if (k < R) row[k] = dsh[k];
      tile_store_rows(a.dL_dshs, a.P, R, rest_tile);
    }
  }
  if (!valid) return;
  a.dL_dmeans3D[3 * (size_t)i] = dmean[0];
  a.dL_dmeans3D[3 * (size_t)i + 1] = dmean[1];
  a.dL_dmeans3D[3 * (size_t)i + 2] = dmean[2];
  a.dL_dmeans2D[3 * (size_t)i] = g0.x;
  a.dL_dmeans2D[3 * (size_t)i + 1] = g0.y;
  a.dL_dmeans2D[3 * (size_t)i + 2] = 0.f;
  a.dL_dopacities[i] = dopac;
  if (a.dL_dcolors) {
    a.dL_dcolors[3 * (size_t)i] = g1.z;
    a.dL_dcolors[3 * (size_t)i + 1] = g1.w;
    a.dL_dcolors[3 * (size_t)i + 2] = g2.x;
  }
  if (a.dL_dscales) {
    a.dL_dscales[3 * (size_t)i] = dscale[0];
    a.dL_dscales[3 * (size_t)i + 1] = dscale[1];
    a.dL_dscales[3 * (size_t)i + 2] = dscale[2];
  }
  if (a.dL_drotations) reinterpret_cast<float4*>(a.dL_drotations)[i] = make_float4(drot[0], drot[1], drot[2], drot[3]);
  if (a.dL_dcov3D) {
#pragma unroll
    for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)i + k] = dc6[k];
  }
}

template <int D>
__global__ void __launch_bounds__(kBlock, 4) preprocess_bwd_kernel(BwdArgs a) {
  extern __shared__ float rest_tile[];   // the block's rows of the SH gradient (raw mode: of _features_rest), stored coalesced
  const int i = blockIdx.x * kBlock + threadIdx.x;
  bwd_one_gaussian<D, false>(a, i, i < a.P, rest_tile);
}

// The outputs are zero already, so only the Gaussians with gradients have anything to do, and in a dense scene
// those are a few per cent.  One wave per workgroup looks at kSparseSpan Gaussians (four per lane: the flags the
// blend backward left, then the rows of sums of the flagged ones), lists those with a non-zero row in LDS, and
// runs the long arithmetic for the listed ones, 64 at a time.  One-wave workgroups because that arithmetic
// takes 128 registers: a wave of it per workgroup of four would leave three quarters of a CU's wave slots
// to workgroups that are already done with their look, and the kernel is a chain of dependent loads per wave.
constexpr int kSparseSpan = 256;

template <int D>
__global__ void __launch_bounds__(64, 4) preprocess_bwd_sparse_kernel(BwdArgs a) {
  __shared__ int s_active[kSparseSpan];
  const int lane = threadIdx.x;
  const int first = blockIdx.x * kSparseSpan + 4 * lane;
  uint32_t flags = 0u;                       // byte k: Gaussian first + k may have a non-zero row
  if (first + 3 < a.P) {
    flags = a.touched ? *reinterpret_cast<const uint32_t*>(a.touched + first) : 0x01010101u;
  } else {
    for (int k = 0; k < 4; k++)
      if (first + k < a.P && (a.touched == nullptr || a.touched[first + k] != 0)) flags |= 1u << (8 * k);
  }
  bool active[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    active[k] = false;
    if ((flags >> (8 * k)) & 0xFFu) {
      const float4* row = reinterpret_cast<const float4*>(a.dL_dsplat) + (size_t)(first + k) * (GSR_GRAD_WORDS / 4);
      const float4 r0 = row[0], r1 = row[1];
      const float r2 = reinterpret_cast<const float*>(row)[8];
      active[k] = (r0.x != 0.f) | (r0.y != 0.f) | (r0.z != 0.f) | (r0.w != 0.f) | (r1.x != 0.f) | (r1.y != 0.f) |
                  (r1.z != 0.f) | (r1.w != 0.f) | (r2 != 0.f);
    }
  }
  int total = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned long long votes = __ballot(active[k]);
    if (active[k]) s_active[total + __popcll(votes & ((1ull << lane) - 1ull))] = first + k;
    total += __popcll(votes);
  }
  __syncthreads();
  for (int at = 0; at < total; at += 64) {
    const bool valid = at + lane < total;
    bwd_one_gaussian<D, true>(a, valid ? s_active[at + lane] : 0, valid, nullptr);
  }
}

int launch_sparse_bwd(int degree, const BwdArgs& a, hipStream_t st) {
  GSR_REQUIRE(((uintptr_t)a.touched & 3) == 0, "preprocess_bwd: touched must be 4-byte aligned");
  const dim3 grid(gsr::div_up(a.P, kSparseSpan)), block(64);
  switch (degree) {
    case 0: hipLaunchKernelGGL(preprocess_bwd_sparse_kernel<0>, grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL(preprocess_bwd_sparse_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(preprocess_bwd_sparse_kernel<2>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(preprocess_bwd_sparse_kernel<3>, grid, block, 0, st, a); break;
  }
  return gsr::check_launch("preprocess_bwd (sparse)");
}

}  // namespace

constexpr int kTotalsHead = 8;   // N, E1, E2, V, status, 3 spare words; the per-block partial sums follow
extern "C" size_t gsr_totals_words(int P) { return kTotalsHead + 4 * (size_t)(P > 0 ? gsr::div_up(P, kBlock) : 0); }

int gsr::totals_partial_blocks(int P) { return P > 0 ? gsr::div_up(P, kBlock) : 0; }
const uint32_t* gsr::totals_partials(const uint32_t* totals) { return totals + kTotalsHead; }

namespace {

int launch_preprocess_fwd(const char* who, FwdArgs a, int D, uint32_t* totals, const uint32_t* status, bool defer_totals,
                          void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int blocks = gsr::div_up(a.P, kBlock);
  dim3 grid(blocks), block(kBlock);
  a.partials = totals + kTotalsHead;
  switch (a.shs ? D : 0) {
    case 0: hipLaunchKernelGGL(preprocess_fwd_kernel<0>, grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL(preprocess_fwd_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(preprocess_fwd_kernel<2>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(preprocess_fwd_kernel<3>, grid, block, 0, st, a); break;
  }
  int rc = gsr::check_launch(who);
  if (rc || defer_totals) return rc;
  return gsr::launch_totals_final(blocks, totals + kTotalsHead, status, totals, st);
}

}  // namespace

int gsr::preprocess_fwd(const PreprocessExtras& extras, int P, int D, int M, const float* means3D, const float* scales,
                        float scale_modifier, const float* rotations, const float* opacities,
                        const float* shs, const float* colors_precomp, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* campos,
                        int W, int H, float tanfovx, float tanfovy, int prefiltered, int reference_lists,
                        uint32_t* records, int32_t* radii, uint32_t* tiles_touched,
                        uint32_t* depth_keys, uint32_t* rects, uint64_t* sort_vals, uint32_t* totals,
                        const uint32_t* status, uint8_t* visible, void* stream) {
  (void)prefiltered;
  GSR_REQUIRE(P >= 0 && W > 0 && H > 0, "preprocess_fwd: bad sizes P=%d W=%d H=%d", P, W, H);
  GSR_REQUIRE(totals && ((uintptr_t)totals & 15) == 0, "preprocess_fwd: totals must be non-null and 16-byte aligned");
  if (P == 0) return gsr::launch_totals_final(0, totals + kTotalsHead, status, totals, (hipStream_t)stream);
  GSR_REQUIRE(means3D && opacities && viewmatrix && projmatrix && campos, "preprocess_fwd: null input");
  GSR_REQUIRE(records && radii && depth_keys && sort_vals, "preprocess_fwd: null output");
  GSR_REQUIRE((shs != nullptr) != (colors_precomp != nullptr), "preprocess_fwd: give exactly one of shs / colors_precomp");
  GSR_REQUIRE(((scales != nullptr) && (rotations != nullptr)) != (cov3D_precomp != nullptr),
              "preprocess_fwd: give exactly one of (scales, rotations) / cov3D_precomp");
  GSR_REQUIRE(D >= 0 && D <= GSR_MAX_SH_DEGREE, "preprocess_fwd: SH degree %d unsupported (0..3)", D);
  GSR_REQUIRE(!shs || M >= (D + 1) * (D + 1), "preprocess_fwd: M=%d too small for degree %d", M, D);
  GSR_REQUIRE(W <= 16 * 16 * 255 && H <= 16 * 16 * 255, "preprocess_fwd: image %dx%d too large", W, H);
  const gsr::Pack pack = gsr::make_pack(P, gsr::div_up(W, GSR_TILE_X), gsr::div_up(H, GSR_TILE_Y));
  GSR_REQUIRE(pack.bits() <= 64, "preprocess_fwd: %d Gaussians on a %dx%d image do not fit the 64-bit sort value", P, W, H);
  FwdArgs a{extras.clear_at, extras.clear_words, 0, 0, reference_lists, P, M, W, H, means3D, scales, rotations, opacities, shs, nullptr, colors_precomp, cov3D_precomp,
            viewmatrix, projmatrix, campos, scale_modifier, tanfovx, tanfovy, pack, records, radii, tiles_touched,
            depth_keys, rects, (unsigned long long*)sort_vals, nullptr, visible};
  return launch_preprocess_fwd("preprocess_fwd", a, D, totals, status, extras.defer_totals, stream);
}

extern "C" int gsr_preprocess_fwd(int P, int D, int M, const float* means3D, const float* scales,
                                  float scale_modifier, const float* rotations, const float* opacities,
                                  const float* shs, const float* colors_precomp, const float* cov3D_precomp,
                                  const float* viewmatrix, const float* projmatrix, const float* campos,
                                  int W, int H, float tanfovx, float tanfovy, int prefiltered, int reference_lists,
                                  uint32_t* records, int32_t* radii, uint32_t* tiles_touched,
                                  uint32_t* depth_keys, uint32_t* rects, uint64_t* sort_vals, uint32_t* totals,
                                  const uint32_t* status, uint8_t* visible, void* stream) {
  return gsr::preprocess_fwd(gsr::PreprocessExtras{false, nullptr, 0}, P, D, M, means3D, scales, scale_modifier, rotations,
                             opacities, shs, colors_precomp, cov3D_precomp, viewmatrix, projmatrix, campos, W, H, tanfovx,
                             tanfovy, prefiltered, reference_lists, records, radii, tiles_touched, depth_keys, rects,
                             sort_vals, totals, status, visible, stream);
}

extern "C" int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                                void* stream) {
  GSR_REQUIRE(P >= 0, "mark_visible: bad P");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(means3D && viewmatrix && present, "mark_visible: null pointer");
  hipLaunchKernelGGL(mark_visible_kernel, dim3(gsr::div_up(P, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     P, means3D, viewmatrix, present);
  return gsr::check_launch("mark_visible");
}

extern "C" int gsr_preprocess_bwd(int P, int D, int M, const float* means3D, const float* scales,
                                  float scale_modifier, const float* rotations, const float* shs,
                                  const float* cov3D_precomp, const float* viewmatrix,
                                  const float* projmatrix, const float* campos, int W, int H, float tanfovx,
                                  float tanfovy, const uint32_t* records, const float* dL_dsplat,
                                  float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacities,
                                  float* dL_dshs, float* dL_dcolors, float* dL_dscales,
                                  float* dL_drotations, float* dL_dcov3D, int outputs_zeroed, const uint8_t* touched,
                                  void* stream) {
  GSR_REQUIRE(P >= 0 && W > 0 && H > 0, "preprocess_bwd: bad sizes");
  GSR_REQUIRE(outputs_zeroed || !touched, "preprocess_bwd: touched goes with outputs_zeroed");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(means3D && viewmatrix && projmatrix && campos && records && dL_dsplat, "preprocess_bwd: null input");
  GSR_REQUIRE(dL_dmeans3D && dL_dmeans2D && dL_dopacities, "preprocess_bwd: null output");
  GSR_REQUIRE((shs != nullptr) == (dL_dshs != nullptr), "preprocess_bwd: shs and dL_dshs go together");
  GSR_REQUIRE((shs != nullptr) != (dL_dcolors != nullptr), "preprocess_bwd: exactly one of dL_dshs / dL_dcolors");
  GSR_REQUIRE(((scales != nullptr) && (rotations != nullptr)) != (cov3D_precomp != nullptr),
              "preprocess_bwd: give exactly one of (scales, rotations) / cov3D_precomp");
  GSR_REQUIRE((cov3D_precomp != nullptr) == (dL_dcov3D != nullptr), "preprocess_bwd: cov3D_precomp and dL_dcov3D go together");
  GSR_REQUIRE((scales != nullptr) == (dL_dscales != nullptr && dL_drotations != nullptr),
              "preprocess_bwd: scales/rotations and their gradients go together");
  GSR_REQUIRE(D >= 0 && D <= GSR_MAX_SH_DEGREE, "preprocess_bwd: SH degree %d unsupported", D);
  GSR_REQUIRE(!shs || (M >= (D + 1) * (D + 1) && M <= 16), "preprocess_bwd: M=%d unsupported for degree %d", M, D);
  BwdArgs a{outputs_zeroed ? 1 : 0, touched, 0, 0, nullptr, nullptr, nullptr, P, M, W, H, means3D, scales, rotations, shs, cov3D_precomp, viewmatrix, projmatrix, campos,
            scale_modifier, tanfovx, tanfovy, records, dL_dsplat, dL_dmeans3D, dL_dmeans2D, dL_dopacities,
            dL_dshs, dL_dcolors, dL_dscales, dL_drotations, dL_dcov3D};
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(gsr::div_up(P, kBlock)), block(kBlock);
  const int d = shs ? D : 0;
  if (a.sparse) return launch_sparse_bwd(d, a, st);
  const size_t lds = shs ? (size_t)kBlock * (M * 3 + 1) * sizeof(float) : 0;     // the block's tile of the SH gradient
  switch (d) {
    case 0: hipLaunchKernelGGL(preprocess_bwd_kernel<0>, grid, block, lds, st, a); break;
    case 1: hipLaunchKernelGGL(preprocess_bwd_kernel<1>, grid, block, lds, st, a); break;
    case 2: hipLaunchKernelGGL(preprocess_bwd_kernel<2>, grid, block, lds, st, a); break;
    default: hipLaunchKernelGGL(preprocess_bwd_kernel<3>, grid, block, lds, st, a); break;
  }
  return gsr::check_launch("preprocess_bwd");
}

// ---------------------------------------------------------------------------
// Raw-parameter variants: the model's exp / sigmoid / normalize / concatenate
// (scene/gaussian_model.py:95-118, five elementwise kernels and a 192 MB copy
// per call at 1M Gaussians, and as many again in backward) folded into the
// per-Gaussian kernels.  Same kernels, `raw` switches the loads and the chain rule.
// ---------------------------------------------------------------------------
int gsr::preprocess_fwd_raw(const PreprocessExtras& extras, int P, int D, int n_rest, const float* xyz,
                            const float* scaling_raw, float scale_modifier, const float* rotation_raw,
                            const float* opacity_raw, const float* features_dc, const float* features_rest,
                            const float* viewmatrix, const float* projmatrix, const float* campos, int W,
                            int H, float tanfovx, float tanfovy, int reference_lists, uint32_t* records,
                            int32_t* radii, uint32_t* tiles_touched, uint32_t* depth_keys, uint32_t* rects,
                            uint64_t* sort_vals, uint32_t* totals, const uint32_t* status,
                            uint8_t* visible, void* stream) {
  GSR_REQUIRE(P >= 0 && W > 0 && H > 0, "preprocess_fwd_raw: bad sizes P=%d W=%d H=%d", P, W, H);
  GSR_REQUIRE(totals && ((uintptr_t)totals & 15) == 0, "preprocess_fwd_raw: totals must be non-null and 16-byte aligned");
  if (P == 0) return gsr::launch_totals_final(0, totals + kTotalsHead, status, totals, (hipStream_t)stream);
  GSR_REQUIRE(xyz && scaling_raw && rotation_raw && opacity_raw && features_dc && viewmatrix && projmatrix && campos,
              "preprocess_fwd_raw: null input");
  GSR_REQUIRE(records && radii && depth_keys && sort_vals, "preprocess_fwd_raw: null output");
  GSR_REQUIRE(D >= 0 && D <= GSR_MAX_SH_DEGREE, "preprocess_fwd_raw: SH degree %d unsupported (0..3)", D);
  GSR_REQUIRE(n_rest >= (D + 1) * (D + 1) - 1 && n_rest <= 15 && (n_rest == 0 || features_rest),
              "preprocess_fwd_raw: n_rest=%d does not hold degree %d", n_rest, D);
  GSR_REQUIRE(W <= 16 * 16 * 255 && H <= 16 * 16 * 255, "preprocess_fwd_raw: image %dx%d too large", W, H);
  const gsr::Pack pack = gsr::make_pack(P, gsr::div_up(W, GSR_TILE_X), gsr::div_up(H, GSR_TILE_Y));
  GSR_REQUIRE(pack.bits() <= 64, "preprocess_fwd_raw: %d Gaussians on a %dx%d image do not fit the 64-bit sort value", P,
              W, H);
  // with no rest tensor the dc tensor alone is a (P,1,3) shs tensor
  FwdArgs a{extras.clear_at, extras.clear_words, 1, n_rest, reference_lists, P, 1 + n_rest, W, H, xyz, scaling_raw, rotation_raw, opacity_raw, features_dc,
            n_rest > 0 ? features_rest : nullptr, nullptr, nullptr, viewmatrix, projmatrix, campos, scale_modifier,
            tanfovx, tanfovy, pack, records, radii, tiles_touched, depth_keys, rects, (unsigned long long*)sort_vals,
            nullptr, visible};
  return launch_preprocess_fwd("preprocess_fwd_raw", a, D, totals, status, extras.defer_totals, stream);
}

extern "C" int gsr_preprocess_fwd_raw(int P, int D, int n_rest, const float* xyz, const float* scaling_raw,
                                      float scale_modifier, const float* rotation_raw, const float* opacity_raw,
                                      const float* features_dc, const float* features_rest,
                                      const float* viewmatrix, const float* projmatrix, const float* campos, int W,
                                      int H, float tanfovx, float tanfovy, int reference_lists, uint32_t* records,
                                      int32_t* radii,
                                      uint32_t* tiles_touched, uint32_t* depth_keys, uint32_t* rects,
                                      uint64_t* sort_vals, uint32_t* totals, const uint32_t* status,
                                      uint8_t* visible, void* stream) {
  return gsr::preprocess_fwd_raw(gsr::PreprocessExtras{false, nullptr, 0}, P, D, n_rest, xyz, scaling_raw, scale_modifier,
                                 rotation_raw, opacity_raw, features_dc, features_rest, viewmatrix, projmatrix, campos, W, H,
                                 tanfovx, tanfovy, reference_lists, records, radii, tiles_touched, depth_keys, rects,
                                 sort_vals, totals, status, visible, stream);
}

extern "C" int gsr_preprocess_bwd_raw(int P, int D, int n_rest, const float* xyz, const float* scaling_raw,
                                      float scale_modifier, const float* rotation_raw, const float* opacity_raw,
                                      const float* features_dc, const float* features_rest,
                                      const float* viewmatrix, const float* projmatrix, const float* campos, int W,
                                      int H, float tanfovx, float tanfovy, const uint32_t* records,
                                      const float* dL_dsplat, float* dL_dxyz, float* dL_dmeans2D,
                                      float* dL_dopacity_raw, float* dL_dfeatures_dc, float* dL_dfeatures_rest,
                                      float* dL_dscaling_raw, float* dL_drotation_raw, int outputs_zeroed,
                                      const uint8_t* touched, void* stream) {
  GSR_REQUIRE(P >= 0 && W > 0 && H > 0, "preprocess_bwd_raw: bad sizes");
  GSR_REQUIRE(outputs_zeroed || !touched, "preprocess_bwd_raw: touched goes with outputs_zeroed");
  if (P == 0) return GSR_OK;
  GSR_REQUIRE(xyz && scaling_raw && rotation_raw && opacity_raw && features_dc && viewmatrix && projmatrix && campos &&
                  records && dL_dsplat, "preprocess_bwd_raw: null input");
  GSR_REQUIRE(dL_dxyz && dL_dmeans2D && dL_dopacity_raw && dL_dfeatures_dc && dL_dscaling_raw && dL_drotation_raw,
              "preprocess_bwd_raw: null output");
  GSR_REQUIRE(D >= 0 && D <= GSR_MAX_SH_DEGREE, "preprocess_bwd_raw: SH degree %d unsupported", D);
  GSR_REQUIRE(n_rest >= (D + 1) * (D + 1) - 1 && n_rest <= 15 && (n_rest == 0 || (features_rest && dL_dfeatures_rest)),
              "preprocess_bwd_raw: n_rest=%d does not hold degree %d", n_rest, D);
  const bool split = n_rest > 0;
  BwdArgs a{outputs_zeroed ? 1 : 0, touched, 1, n_rest, split ? features_rest : nullptr, opacity_raw, split ? dL_dfeatures_rest : nullptr,
            P, 1 + n_rest, W, H, xyz, scaling_raw, rotation_raw, features_dc, nullptr, viewmatrix, projmatrix, campos,
            scale_modifier, tanfovx, tanfovy, records, dL_dsplat, dL_dxyz, dL_dmeans2D, dL_dopacity_raw,
            dL_dfeatures_dc, nullptr, dL_dscaling_raw, dL_drotation_raw, nullptr};
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(gsr::div_up(P, kBlock)), block(kBlock);
  // the block's tile of the SH gradient: the rows of features_rest, or of features_dc when there is no rest tensor
  if (a.sparse) return launch_sparse_bwd(D, a, st);
  const size_t lds = (size_t)kBlock * (split ? n_rest * 3 : 3 + 1) * sizeof(float);
  switch (D) {
    case 0: hipLaunchKernelGGL(preprocess_bwd_kernel<0>, grid, block, lds, st, a); break;
    case 1: hipLaunchKernelGGL(preprocess_bwd_kernel<1>, grid, block, lds, st, a); break;
    case 2: hipLaunchKernelGGL(preprocess_bwd_kernel<2>, grid, block, lds, st, a); break;
    default: hipLaunchKernelGGL(preprocess_bwd_kernel<3>, grid, block, lds, st, a); break;
  }
  return gsr::check_launch("preprocess_bwd_raw");
}
