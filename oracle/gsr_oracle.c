/*
 * CPU ORACLE of the tile-based differentiable Gaussian rasterizer.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * and the benchmark's cpu_baseline leg may load it; the shipped path
 * (gaussian-splatting-own_amd/) never does.
 *
 * PARITY UNPINNED at the rasterizer boundary: the reference calls
 * diff_gaussian_rasterization (gaussian_renderer/__init__.py:14,36-51,85-93),
 * an un-vendored, un-pinned submodule (.gitmodules:4-6, empty directory) and
 * holds no test, golden vector or fixture for it.  This file restates the
 * published algorithm of that module (3D Gaussian Splatting, Kerbl et al. 2023;
 * graphdeco-inria/diff-gaussian-rasterization, two-output API generation):
 * preprocess -> per-tile key duplication -> stable sort by (tile, depth) ->
 * tile ranges -> front-to-back alpha blending, and the backward of all of it.
 * What IS pinned by the reference's own Python, and checked by the tests
 * against golden vectors generated from it:
 *   - SH -> RGB:   utils/sh_utils.py:57-112, gaussian_renderer/__init__.py:74-78
 *   - covariance:  scene/gaussian_model.py:27-31, utils/general_utils.py:64-110
 *   - camera conventions (row-vector, transposed matrices):
 *                  scene/cameras.py:54-57, utils/graphics_utils.py:22-71
 *   - tensor layouts: scene/gaussian_model.py:95-118 (shs are (P,16,3))
 *
 * Built twice from this one source: REAL=float (the parity oracle; compiled
 * with -ffp-contract=off so every + - * / sqrt is a single IEEE operation, the
 * same sequence the HIP preprocess kernel executes) and REAL=double (used for
 * gradient checks only).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gsr_constants.h"

#ifndef REAL
#define REAL float
#endif

#define K(x) ((REAL)(x))

static inline REAL r_sqrt(REAL x) { return sizeof(REAL) == 4 ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }
static inline REAL r_exp(REAL x) { return sizeof(REAL) == 4 ? (REAL)expf((float)x) : (REAL)exp((double)x); }
static inline REAL r_ceil(REAL x) { return sizeof(REAL) == 4 ? (REAL)ceilf((float)x) : (REAL)ceil((double)x); }
static inline REAL r_exp2(REAL x) { return sizeof(REAL) == 4 ? (REAL)exp2f((float)x) : (REAL)exp2((double)x); }
static inline REAL r_fma(REAL a, REAL b, REAL c) { return sizeof(REAL) == 4 ? (REAL)fmaf((float)a, (float)b, (float)c) : (REAL)fma((double)a, (double)b, (double)c); }
static inline REAL r_min(REAL a, REAL b) { return a < b ? a : b; }
static inline REAL r_max(REAL a, REAL b) { return a > b ? a : b; }
static inline REAL r_abs(REAL a) { return a < 0 ? -a : a; }

/* Camera + raster settings: the scalar and small-matrix fields of
 * GaussianRasterizationSettings (gaussian_renderer/__init__.py:36-49). */
typedef struct {
  int32_t P;           /* number of Gaussians                                     */
  int32_t D;           /* active SH degree (sh_degree)                            */
  int32_t M;           /* SH coefficients stored per Gaussian and channel (16)    */
  int32_t W, H;        /* image_width, image_height                               */
  int32_t prefiltered; /* always 0 in the reference (:47)                         */
  REAL tanfovx, tanfovy, scale_modifier;
  REAL view[16];       /* world_view_transform, row-major as stored (cameras.py:54) */
  REAL proj[16];       /* full_proj_transform (cameras.py:56)                     */
  REAL campos[3];      /* camera_center (cameras.py:57)                           */
  REAL bg[3];          /* bg                                                      */
} gso_cam;

int gso_real_bytes(void) { return (int)sizeof(REAL); }

/* The Gaussian's exponent and value at offset (dx, dy) from its centre.
 * Default build: the expression as the published algorithm writes it,
 * power = -1/2 (A dx^2 + C dy^2) - B dx dy, one IEEE operation after the other,
 * G = exp(power).
 * GSO_KERNEL_ORDER build (a second fp32 oracle, used to show that what separates
 * the HIP blend kernels from the default oracle on ill-conditioned conics is the
 * rounding of this one expression and nothing else): the sequence of the HIP
 * kernels' gauss_exponent(), i.e. the conic pre-scaled into the log2 domain,
 * (la, lb, lc) = (-log2e/2 A, -log2e B, -log2e/2 C) each rounded once, then
 * p2 = fma(lb dx, dy, fma(la dx, dx, (lc dy) dy)) and G = exp2(p2).
 * Returns G; *power receives the exponent in the natural-log domain (its sign is
 * the skip test, its size feeds the margin estimate). */
int gso_kernel_order(void) {
#ifdef GSO_KERNEL_ORDER
  return 1;
#else
  return 0;
#endif
}
static inline REAL gauss_value(REAL A, REAL B, REAL Cc, REAL dx, REAL dy, REAL* power) {
#ifdef GSO_KERNEL_ORDER
  const REAL log2e = K(1.4426950408889634);
  const REAL la = (K(-0.5) * log2e) * A, lb = (-log2e) * B, lc = (K(-0.5) * log2e) * Cc;
  const REAL p2 = r_fma(dx, r_fma(la, dx, lb * dy), (lc * dy) * dy);
  *power = p2 * K(0.6931471805599453);
  return r_exp2(p2);
#else
  *power = K(-0.5) * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
  return r_exp(*power);
#endif
}

/* p' = [x y z 1] * M for a (4,4) row-major M: the reference stores its
 * matrices transposed and multiplies row vectors (graphics_utils.py:22-29). */
static inline void xform43(const REAL* m, const REAL* p, REAL* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform44(const REAL* m, const REAL* p, REAL* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Rotation matrix of a quaternion (w,x,y,z) taken as given, NOT normalised
 * here: the reference feeds get_rotation = normalize(q)
 * (scene/gaussian_model.py:100-101); formula as utils/general_utils.py:90-98. */
static inline void quat_to_rot(const REAL* q, REAL* R) {
  REAL r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = K(1.0) - K(2.0) * (y * y + z * z);
  R[1] = K(2.0) * (x * y - r * z);
  R[2] = K(2.0) * (x * z + r * y);
  R[3] = K(2.0) * (x * y + r * z);
  R[4] = K(1.0) - K(2.0) * (x * x + z * z);
  R[5] = K(2.0) * (y * z - r * x);
  R[6] = K(2.0) * (x * z - r * y);
  R[7] = K(2.0) * (y * z + r * x);
  R[8] = K(1.0) - K(2.0) * (x * x + y * y);
}

/* Sigma = L L^T, L = R diag(mod*s); packed (xx,xy,xz,yy,yz,zz)
 * (scene/gaussian_model.py:27-31, utils/general_utils.py:64-73,101-110). */
static inline void cov3d_from_scale_rot(const REAL* s, REAL mod, const REAL* q, REAL* c6) {
  REAL R[9], L[9];
  quat_to_rot(q, R);
  REAL sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
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

/* Intermediate values of the EWA projection, shared by forward and backward. */
typedef struct {
  REAL tx, ty, tz;      /* view-space mean with x/z, y/z clamped             */
  REAL xmul, ymul;      /* 0 where the clamp was active, else 1               */
  REAL M0[3], M1[3];    /* rows of J*W (2x3)                                  */
  REAL a, b, c;         /* 2D covariance incl. dilation                       */
} ewa_t;

static inline void ewa_project(const gso_cam* cam, const REAL* pview, const REAL* c6, ewa_t* e) {
  const REAL* V = cam->view;
  REAL fx = (REAL)cam->W / (K(2.0) * cam->tanfovx);
  REAL fy = (REAL)cam->H / (K(2.0) * cam->tanfovy);
  REAL limx = K(GSR_FRUSTUM_CLAMP) * cam->tanfovx;
  REAL limy = K(GSR_FRUSTUM_CLAMP) * cam->tanfovy;
  REAL tz = pview[2];
  REAL txtz = pview[0] / tz, tytz = pview[1] / tz;
  e->xmul = (txtz < -limx || txtz > limx) ? K(0.0) : K(1.0);
  e->ymul = (tytz < -limy || tytz > limy) ? K(0.0) : K(1.0);
  REAL tx = r_min(limx, r_max(-limx, txtz)) * tz;
  REAL ty = r_min(limy, r_max(-limy, tytz)) * tz;
  e->tx = tx; e->ty = ty; e->tz = tz;
  REAL J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
  REAL J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
  /* W[k][j] = V[4j+k]: world axis j -> view axis k */
  for (int j = 0; j < 3; j++) {
    REAL W0 = V[4 * j + 0], W1 = V[4 * j + 1], W2 = V[4 * j + 2];
    e->M0[j] = J00 * W0 + J02 * W2;
    e->M1[j] = J11 * W1 + J12 * W2;
  }
  const REAL* M0 = e->M0; const REAL* M1 = e->M1;
  REAL v0[3], v1[3];
  v0[0] = c6[0] * M0[0] + c6[1] * M0[1] + c6[2] * M0[2];
  v0[1] = c6[1] * M0[0] + c6[3] * M0[1] + c6[4] * M0[2];
  v0[2] = c6[2] * M0[0] + c6[4] * M0[1] + c6[5] * M0[2];
  v1[0] = c6[0] * M1[0] + c6[1] * M1[1] + c6[2] * M1[2];
  v1[1] = c6[1] * M1[0] + c6[3] * M1[1] + c6[4] * M1[2];
  v1[2] = c6[2] * M1[0] + c6[4] * M1[1] + c6[5] * M1[2];
  e->a = (M0[0] * v0[0] + M0[1] * v0[1] + M0[2] * v0[2]) + K(GSR_COV2D_DILATION);
  e->b = M0[0] * v1[0] + M0[1] * v1[1] + M0[2] * v1[2];
  e->c = (M1[0] * v1[0] + M1[1] * v1[1] + M1[2] * v1[2]) + K(GSR_COV2D_DILATION);
}

/* The 16 real-SH basis values of a unit direction (utils/sh_utils.py:74-100);
 * b[k] multiplies sh[k]. */
static inline void sh_basis(int D, REAL x, REAL y, REAL z, REAL* b) {
  b[0] = K(GSR_SH_C0);
  if (D > 0) {
    b[1] = -K(GSR_SH_C1) * y;
    b[2] = K(GSR_SH_C1) * z;
    b[3] = -K(GSR_SH_C1) * x;
    if (D > 1) {
      REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = K(GSR_SH_C2_0) * xy;
      b[5] = K(GSR_SH_C2_1) * yz;
      b[6] = K(GSR_SH_C2_2) * (K(2.0) * zz - xx - yy);
      b[7] = K(GSR_SH_C2_3) * xz;
      b[8] = K(GSR_SH_C2_4) * (xx - yy);
      if (D > 2) {
        b[9] = K(GSR_SH_C3_0) * y * (K(3.0) * xx - yy);
        b[10] = K(GSR_SH_C3_1) * xy * z;
        b[11] = K(GSR_SH_C3_2) * y * (K(4.0) * zz - xx - yy);
        b[12] = K(GSR_SH_C3_3) * z * (K(2.0) * zz - K(3.0) * xx - K(3.0) * yy);
        b[13] = K(GSR_SH_C3_4) * x * (K(4.0) * zz - xx - yy);
        b[14] = K(GSR_SH_C3_5) * z * (xx - yy);
        b[15] = K(GSR_SH_C3_6) * x * (xx - K(3.0) * yy);
      }
    }
  }
}

static inline int ncoef(int D) { return (D + 1) * (D + 1); }

/* Tile rectangle [x0,x1) x [y0,y1) touched by a splat of integer radius r at
 * pixel position (px,py): conservative bounding square, clamped to the grid. */
static inline void tile_rect(REAL px, REAL py, int r, int gx, int gy, int* rc) {
  REAL rf = (REAL)r;
  REAL x0 = (px - rf) / K(GSR_TILE_X), x1 = (px + rf + K(GSR_TILE_X - 1)) / K(GSR_TILE_X);
  REAL y0 = (py - rf) / K(GSR_TILE_Y), y1 = (py + rf + K(GSR_TILE_Y - 1)) / K(GSR_TILE_Y);
  /* truncation toward zero after clamping into [0, grid] == clamp(trunc()) */
  rc[0] = (int)r_min((REAL)gx, r_max(K(0.0), x0));
  rc[1] = (int)r_min((REAL)gy, r_max(K(0.0), y0));
  rc[2] = (int)r_min((REAL)gx, r_max(K(0.0), x1));
  rc[3] = (int)r_min((REAL)gy, r_max(K(0.0), y1));
}

/* ------------------------------------------------------------------------ */
/* Stage 1: preprocess forward (one Gaussian at a time).                     */
/* Outputs for culled Gaussians: radii = 0, tiles_touched = 0, the rest      */
/* untouched (caller zero-fills).                                            */
/* ------------------------------------------------------------------------ */
void gso_preprocess_fwd(const gso_cam* cam, const REAL* means3D, const REAL* scales,
                        const REAL* rotations, const REAL* opacities, const REAL* shs,
                        const REAL* colors_precomp, const REAL* cov3D_precomp,
                        int32_t* radii, REAL* means2D, REAL* depths, REAL* cov3D, REAL* rgb,
                        REAL* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped,
                        int32_t* rects) {
  const int P = cam->P;
  const int gx = (cam->W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (cam->H + GSR_TILE_Y - 1) / GSR_TILE_Y;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    radii[i] = 0;
    tiles_touched[i] = 0;
    const REAL* p = means3D + 3 * i;
    REAL pv[3], ph[4];
    xform43(cam->view, p, pv);
    if (pv[2] <= K(GSR_NEAR_CULL)) continue;
    xform44(cam->proj, p, ph);
    REAL pw = K(1.0) / (ph[3] + K(GSR_W_EPS));
    REAL ndcx = ph[0] * pw, ndcy = ph[1] * pw;

    REAL* c6 = cov3D + 6 * i;
    if (cov3D_precomp) {
      for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * i + k];
    } else {
      cov3d_from_scale_rot(scales + 3 * i, cam->scale_modifier, rotations + 4 * i, c6);
    }
    ewa_t e;
    ewa_project(cam, pv, c6, &e);
    REAL det = e.a * e.c - e.b * e.b;
    if (det == K(0.0)) continue;
    REAL det_inv = K(1.0) / det;
    REAL conA = e.c * det_inv, conB = -e.b * det_inv, conC = e.a * det_inv;
    REAL mid = K(0.5) * (e.a + e.c);
    REAL disc = r_sqrt(r_max(K(GSR_LAMBDA_FLOOR), mid * mid - det));
    REAL lam1 = mid + disc, lam2 = mid - disc;
    REAL rf = r_ceil(K(GSR_RADIUS_SIGMAS) * r_sqrt(r_max(lam1, lam2)));
    int r = (rf < K(2147483520.0)) ? (int)rf : 2147483647;
    REAL px = ((ndcx + K(1.0)) * (REAL)cam->W - K(1.0)) * K(0.5);
    REAL py = ((ndcy + K(1.0)) * (REAL)cam->H - K(1.0)) * K(0.5);
    int rc[4];
    tile_rect(px, py, r, gx, gy, rc);
    int area = (rc[2] - rc[0]) * (rc[3] - rc[1]);
    if (area <= 0) continue;

    if (colors_precomp) {
      for (int ch = 0; ch < 3; ch++) rgb[3 * i + ch] = colors_precomp[3 * i + ch];
      clamped[3 * i] = clamped[3 * i + 1] = clamped[3 * i + 2] = 0;
    } else {
      /* gaussian_renderer/__init__.py:75-78 + utils/sh_utils.py:57-112 */
      REAL dx = p[0] - cam->campos[0], dy = p[1] - cam->campos[1], dz = p[2] - cam->campos[2];
      REAL len = r_sqrt(dx * dx + dy * dy + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      REAL b[16];
      sh_basis(cam->D, dx, dy, dz, b);
      const REAL* sh = shs + (size_t)i * cam->M * 3;
      int n = ncoef(cam->D);
      for (int ch = 0; ch < 3; ch++) {
        REAL acc = b[0] * sh[ch];
        for (int k = 1; k < n; k++) acc = acc + b[k] * sh[3 * k + ch];
        acc = acc + K(GSR_SH_OFFSET);
        clamped[3 * i + ch] = (acc < K(0.0));
        rgb[3 * i + ch] = r_max(acc, K(0.0));
      }
    }
    depths[i] = pv[2];
    radii[i] = r;
    means2D[2 * i] = px;
    means2D[2 * i + 1] = py;
    conic_opacity[4 * i + 0] = conA;
    conic_opacity[4 * i + 1] = conB;
    conic_opacity[4 * i + 2] = conC;
    conic_opacity[4 * i + 3] = opacities[i];
    tiles_touched[i] = (uint32_t)area;
    if (rects) { for (int k = 0; k < 4; k++) rects[4 * i + k] = rc[k]; }
  }
}

/* Frustum test alone (markVisible of the rasterizer module; the reference's
 * Python never calls it). */
void gso_mark_visible(const gso_cam* cam, const REAL* means3D, uint8_t* present) {
  for (int i = 0; i < cam->P; i++) {
    REAL pv[3];
    xform43(cam->view, means3D + 3 * i, pv);
    present[i] = pv[2] > K(GSR_NEAR_CULL);
  }
}

/* ------------------------------------------------------------------------ */
/* Stage 2: binning.  One (key,value) per touched tile, key = tile id in the  */
/* high 32 bits and the IEEE bits of the fp32 depth in the low 32 bits,       */
/* emitted tile-row-major per Gaussian in ascending Gaussian index, then a    */
/* STABLE ascending sort by key: ties keep ascending Gaussian index.          */
/* ------------------------------------------------------------------------ */
static void radix_sort_pairs_u64(uint64_t* keys, uint32_t* vals, uint64_t* k2, uint32_t* v2,
                                 size_t n, int bits) {
  uint64_t* ks = keys; uint32_t* vs = vals; uint64_t* kd = k2; uint32_t* vd = v2;
  int npass = (bits + 7) / 8;
  for (int pass = 0; pass < npass; pass++) {
    size_t cnt[257];
    memset(cnt, 0, sizeof(cnt));
    int sh = pass * 8;
    for (size_t i = 0; i < n; i++) cnt[((ks[i] >> sh) & 0xff) + 1]++;
    for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
    for (size_t i = 0; i < n; i++) {
      size_t dst = cnt[(ks[i] >> sh) & 0xff]++;
      kd[dst] = ks[i]; vd[dst] = vs[i];
    }
    uint64_t* tk = ks; ks = kd; kd = tk;
    uint32_t* tv = vs; vs = vd; vd = tv;
  }
  if (ks != keys) { memcpy(keys, ks, n * sizeof(uint64_t)); memcpy(vals, vs, n * sizeof(uint32_t)); }
}

/* depths32 are always fp32 (the key holds fp32 bits in both builds).
 * offsets = inclusive prefix sum of tiles_touched.  ranges = (start,end) per
 * tile, (0,0) for empty tiles. Returns 0 on success. */
int gso_bin(const gso_cam* cam, const int32_t* radii, const REAL* means2D, const float* depths32,
            const uint32_t* offsets, uint64_t n_dup, uint64_t* keys_unsorted,
            uint32_t* vals_unsorted, uint64_t* keys_sorted, uint32_t* vals_sorted,
            uint32_t* ranges) {
  const int P = cam->P;
  const int gx = (cam->W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (cam->H + GSR_TILE_Y - 1) / GSR_TILE_Y;
  for (int i = 0; i < P; i++) {
    if (radii[i] <= 0) continue;
    uint64_t off = i == 0 ? 0 : offsets[i - 1];
    int rc[4];
    tile_rect(means2D[2 * i], means2D[2 * i + 1], radii[i], gx, gy, rc);
    uint32_t dbits;
    memcpy(&dbits, depths32 + i, 4);
    for (int y = rc[1]; y < rc[3]; y++)
      for (int x = rc[0]; x < rc[2]; x++) {
        if (off >= n_dup) return -1;
        keys_sorted[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
        vals_sorted[off] = (uint32_t)i;
        off++;
      }
    if (off != offsets[i]) return -2;
  }
  if (keys_unsorted) memcpy(keys_unsorted, keys_sorted, n_dup * sizeof(uint64_t));
  if (vals_unsorted) memcpy(vals_unsorted, vals_sorted, n_dup * sizeof(uint32_t));
  uint64_t* k2 = (uint64_t*)malloc((n_dup + 1) * sizeof(uint64_t));
  uint32_t* v2 = (uint32_t*)malloc((n_dup + 1) * sizeof(uint32_t));
  if (!k2 || !v2) { free(k2); free(v2); return -3; }
  int tile_bits = 1;
  while ((1u << tile_bits) < (unsigned)(gx * gy)) tile_bits++;
  radix_sort_pairs_u64(keys_sorted, vals_sorted, k2, v2, n_dup, 32 + tile_bits);
  free(k2); free(v2);
  memset(ranges, 0, (size_t)gx * gy * 2 * sizeof(uint32_t));
  for (uint64_t i = 0; i < n_dup; i++) {
    uint32_t t = (uint32_t)(keys_sorted[i] >> 32);
    if (i == 0) ranges[2 * t] = 0;
    else {
      uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
      if (t != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * t] = (uint32_t)i; }
    }
    if (i == n_dup - 1) ranges[2 * t + 1] = (uint32_t)n_dup;
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Stage 3: blend forward.  Pixel centres are integer coordinates.           */
/* margin (optional, per pixel): min over every threshold decision taken for */
/* the pixel of |distance to the threshold| / (fp32 rounding uncertainty of  */
/* the compared quantity).  A pixel with margin < ~4 may legitimately differ */
/* between two fp32 evaluations of the same formulas (different exp          */
/* implementation, FMA contraction), by up to one skipped contribution.      */
/* ------------------------------------------------------------------------ */
void gso_blend_fwd(const gso_cam* cam, const uint32_t* ranges, const uint32_t* point_list,
                   const REAL* means2D, const REAL* colors, const REAL* conic_opacity,
                   REAL* out_color, REAL* final_T, uint32_t* n_contrib, float* margin,
                   uint64_t* n_evaluated) {
  const int W = cam->W, H = cam->H;
  const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
  uint64_t evals = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : evals)
  for (int tile = 0; tile < gx * gy; tile++) {
    const uint32_t start = ranges[2 * tile], end = ranges[2 * tile + 1];
    const int tx0 = (tile % gx) * GSR_TILE_X, ty0 = (tile / gx) * GSR_TILE_Y;
    for (int ly = 0; ly < GSR_TILE_Y; ly++)
      for (int lx = 0; lx < GSR_TILE_X; lx++) {
        const int pxi = tx0 + lx, pyi = ty0 + ly;
        if (pxi >= W || pyi >= H) continue;
        const REAL pxf = (REAL)pxi, pyf = (REAL)pyi;
        REAL T = K(1.0), C0 = 0, C1 = 0, C2 = 0;
        uint32_t contributor = 0, last = 0;
        float mg = 1e30f, relT = 0.f;
        for (uint32_t s = start; s < end; s++) {
          contributor++;
          const uint32_t g = point_list[s];
          const REAL dx = means2D[2 * g] - pxf, dy = means2D[2 * g + 1] - pyf;
          const REAL A = conic_opacity[4 * g], B = conic_opacity[4 * g + 1],
                     Cc = conic_opacity[4 * g + 2], o = conic_opacity[4 * g + 3];
          REAL power;
          const REAL G = gauss_value(A, B, Cc, dx, dy, &power);
          evals++;
          float perr = 0.f;
          if (margin) {
            float mag = 0.5f * (float)(r_abs(A) * dx * dx + r_abs(Cc) * dy * dy) + (float)r_abs(B * dx * dy);
            perr = 6e-7f * mag;
            float m = (float)r_abs(power) / (perr + 1e-30f);
            if (m < mg) mg = m;
          }
          if (power > K(0.0)) continue;
          const REAL alpha = r_min(K(GSR_ALPHA_MAX), o * G);
          float rela = perr + 4e-7f;
          if (margin) {
            float m = fabsf((float)alpha - GSR_ALPHA_MIN) / ((float)alpha * rela + 1e-30f);
            if (m < mg) mg = m;
          }
          if (alpha < K(GSR_ALPHA_MIN)) continue;
          const REAL test_T = T * (K(1.0) - alpha);
          if (margin) {
            relT += (float)alpha * rela / (1.f - (float)alpha) + 1.2e-7f;
            float m = fabsf((float)test_T - GSR_T_STOP) / ((float)test_T * relT + 1e-30f);
            if (m < mg) mg = m;
          }
          if (test_T < K(GSR_T_STOP)) break;
          const REAL w = alpha * T;
          C0 += colors[3 * g] * w; C1 += colors[3 * g + 1] * w; C2 += colors[3 * g + 2] * w;
          T = test_T;
          last = contributor;
        }
        const size_t pix = (size_t)pyi * W + pxi;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * cam->bg[0];
        out_color[(size_t)H * W + pix] = C1 + T * cam->bg[1];
        out_color[(size_t)2 * H * W + pix] = C2 + T * cam->bg[2];
        if (margin) margin[pix] = mg;
      }
  }
  if (n_evaluated) *n_evaluated = evals;
}

/* ------------------------------------------------------------------------ */
/* Stage 4: blend backward.  Per pixel, back to front from its last           */
/* contributor, with the transmittance unwound by T <- T / (1 - alpha).       */
/* Sums over pixels are taken in double whatever REAL is.                     */
/* Outputs (zero-filled here): dL_dmean2D (P,2) in NDC-scaled units (x by W/2,*/
/* y by H/2: this is what lands in .grad of the means2D sink,                 */
/* gaussian_renderer/__init__.py:26-30, scene/gaussian_model.py:405-407);     */
/* dL_dconic (P,3) = d/d(A, B, C) with B the off-diagonal entry counted once  */
/* (power = -1/2 (A dx^2 + C dy^2) - B dx dy); dL_dopacity (P); dL_dcolor(P,3)*/
/* The alpha cap (min with 0.99) is treated as identity, as the module does.  */
/* ------------------------------------------------------------------------ */
void gso_blend_bwd(const gso_cam* cam, const uint32_t* ranges, const uint32_t* point_list,
                   const REAL* means2D, const REAL* colors, const REAL* conic_opacity,
                   const REAL* final_T, const uint32_t* n_contrib, const REAL* dL_dpix,
                   REAL* dL_dmean2D, REAL* dL_dconic, REAL* dL_dopacity, REAL* dL_dcolor) {
  const int W = cam->W, H = cam->H, P = cam->P;
  const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
  double* acc = (double*)calloc((size_t)P * 9, sizeof(double));
  const REAL ddelx_dx = K(0.5) * (REAL)W, ddely_dy = K(0.5) * (REAL)H;
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < gx * gy; tile++) {
    const uint32_t start = ranges[2 * tile], end = ranges[2 * tile + 1];
    const int tx0 = (tile % gx) * GSR_TILE_X, ty0 = (tile / gx) * GSR_TILE_Y;
    for (int ly = 0; ly < GSR_TILE_Y; ly++)
      for (int lx = 0; lx < GSR_TILE_X; lx++) {
        const int pxi = tx0 + lx, pyi = ty0 + ly;
        if (pxi >= W || pyi >= H) continue;
        const size_t pix = (size_t)pyi * W + pxi;
        const REAL pxf = (REAL)pxi, pyf = (REAL)pyi;
        const REAL Tf = final_T[pix];
        REAL T = Tf;
        const uint32_t last = n_contrib[pix];
        REAL dpix[3] = {dL_dpix[pix], dL_dpix[(size_t)H * W + pix], dL_dpix[(size_t)2 * H * W + pix]};
        REAL accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
        const REAL bg_dot = cam->bg[0] * dpix[0] + cam->bg[1] * dpix[1] + cam->bg[2] * dpix[2];
        for (uint32_t k = last; k > 0; k--) {
          const uint32_t s = start + k - 1;
          if (s >= end) continue;
          const uint32_t g = point_list[s];
          const REAL dx = means2D[2 * g] - pxf, dy = means2D[2 * g + 1] - pyf;
          const REAL A = conic_opacity[4 * g], B = conic_opacity[4 * g + 1],
                     Cc = conic_opacity[4 * g + 2], o = conic_opacity[4 * g + 3];
          REAL power;
          const REAL G = gauss_value(A, B, Cc, dx, dy, &power);
          if (power > K(0.0)) continue;
          const REAL alpha = r_min(K(GSR_ALPHA_MAX), o * G);
          if (alpha < K(GSR_ALPHA_MIN)) continue;
          T = T / (K(1.0) - alpha);
          const REAL dch_dcolor = alpha * T;
          REAL dL_dalpha = 0;
          double* a9 = acc + (size_t)g * 9;
          for (int ch = 0; ch < 3; ch++) {
            const REAL c = colors[3 * g + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (K(1.0) - last_alpha) * accum_rec[ch];
            last_color[ch] = c;
            dL_dalpha += (c - accum_rec[ch]) * dpix[ch];
#pragma omp atomic
            a9[6 + ch] += (double)(dch_dcolor * dpix[ch]);
          }
          dL_dalpha *= T;
          last_alpha = alpha;
          dL_dalpha += (-Tf / (K(1.0) - alpha)) * bg_dot;
          const REAL dL_dG = o * dL_dalpha;
          const REAL gdx = G * dx, gdy = G * dy;
          const REAL dG_ddelx = -gdx * A - gdy * B;
          const REAL dG_ddely = -gdy * Cc - gdx * B;
#pragma omp atomic
          a9[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
#pragma omp atomic
          a9[1] += (double)(dL_dG * dG_ddely * ddely_dy);
#pragma omp atomic
          a9[2] += (double)(K(-0.5) * gdx * dx * dL_dG);
#pragma omp atomic
          a9[3] += (double)(-gdx * dy * dL_dG);
#pragma omp atomic
          a9[4] += (double)(K(-0.5) * gdy * dy * dL_dG);
#pragma omp atomic
          a9[5] += (double)(G * dL_dalpha);
        }
      }
  }
  for (int i = 0; i < P; i++) {
    const double* a9 = acc + (size_t)i * 9;
    dL_dmean2D[2 * i] = (REAL)a9[0]; dL_dmean2D[2 * i + 1] = (REAL)a9[1];
    dL_dconic[3 * i] = (REAL)a9[2]; dL_dconic[3 * i + 1] = (REAL)a9[3]; dL_dconic[3 * i + 2] = (REAL)a9[4];
    dL_dopacity[i] = (REAL)a9[5];
    dL_dcolor[3 * i] = (REAL)a9[6]; dL_dcolor[3 * i + 1] = (REAL)a9[7]; dL_dcolor[3 * i + 2] = (REAL)a9[8];
  }
  free(acc);
}

/* ------------------------------------------------------------------------ */
/* Stage 5: preprocess backward.  Only Gaussians with radii > 0 receive       */
/* gradients.  Outputs are zero-filled here.                                  */
/*   dL_dmeans3D (P,3), dL_dcov3D (P,6) [gradient w.r.t. the 6 packed entries,*/
/*   an off-diagonal entry standing for both symmetric positions],            */
/*   dL_dshs (P,M,3), dL_dscales (P,3), dL_drotations (P,4).                  */
/* When colors_precomp / cov3D_precomp were used, the SH / scale-rotation     */
/* parts are skipped (shs == NULL / scales == NULL).                          */
/* ------------------------------------------------------------------------ */
void gso_preprocess_bwd(const gso_cam* cam, const int32_t* radii, const REAL* means3D,
                        const REAL* scales, const REAL* rotations, const REAL* shs,
                        const uint8_t* clamped, const REAL* cov3D, const REAL* dL_dmean2D,
                        const REAL* dL_dconic, const REAL* dL_dcolor, REAL* dL_dmeans3D,
                        REAL* dL_dcov3D, REAL* dL_dshs, REAL* dL_dscales, REAL* dL_drotations) {
  const int P = cam->P;
  memset(dL_dmeans3D, 0, (size_t)P * 3 * sizeof(REAL));
  memset(dL_dcov3D, 0, (size_t)P * 6 * sizeof(REAL));
  if (dL_dshs) memset(dL_dshs, 0, (size_t)P * cam->M * 3 * sizeof(REAL));
  if (dL_dscales) memset(dL_dscales, 0, (size_t)P * 3 * sizeof(REAL));
  if (dL_drotations) memset(dL_drotations, 0, (size_t)P * 4 * sizeof(REAL));
  const REAL* V = cam->view;
  const REAL* PV = cam->proj;
  const REAL fx = (REAL)cam->W / (K(2.0) * cam->tanfovx);
  const REAL fy = (REAL)cam->H / (K(2.0) * cam->tanfovy);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    if (radii[i] <= 0) continue;
    const REAL* p = means3D + 3 * i;
    const REAL* c6 = cov3D + 6 * i;
    REAL dmean[3] = {0, 0, 0};

    /* (a) conic -> cov2D -> cov3D and view-space mean */
    REAL pv[3];
    xform43(V, p, pv);
    ewa_t e;
    ewa_project(cam, pv, c6, &e);
    const REAL a = e.a, b = e.b, c = e.c;
    const REAL dA = dL_dconic[3 * i], dB = dL_dconic[3 * i + 1], dC = dL_dconic[3 * i + 2];
    const REAL det = a * c - b * b;
    const REAL det2inv = K(1.0) / (det * det + K(GSR_DET2_EPS));
    REAL dL_da = 0, dL_db = 0, dL_dc = 0;
    if (det2inv != K(0.0)) {
      dL_da = det2inv * (-c * c * dA + b * c * dB + (det - a * c) * dC);
      dL_dc = det2inv * (-a * a * dC + a * b * dB + (det - a * c) * dA);
      dL_db = det2inv * (K(2.0) * b * c * dA - (det + K(2.0) * b * b) * dB + K(2.0) * a * b * dC);
    }
    const REAL* M0 = e.M0; const REAL* M1 = e.M1;
    REAL* dc6 = dL_dcov3D + 6 * i;
    dc6[0] = M0[0] * M0[0] * dL_da + M0[0] * M1[0] * dL_db + M1[0] * M1[0] * dL_dc;
    dc6[3] = M0[1] * M0[1] * dL_da + M0[1] * M1[1] * dL_db + M1[1] * M1[1] * dL_dc;
    dc6[5] = M0[2] * M0[2] * dL_da + M0[2] * M1[2] * dL_db + M1[2] * M1[2] * dL_dc;
    dc6[1] = K(2.0) * M0[0] * M0[1] * dL_da + (M0[0] * M1[1] + M0[1] * M1[0]) * dL_db + K(2.0) * M1[0] * M1[1] * dL_dc;
    dc6[2] = K(2.0) * M0[0] * M0[2] * dL_da + (M0[0] * M1[2] + M0[2] * M1[0]) * dL_db + K(2.0) * M1[0] * M1[2] * dL_dc;
    dc6[4] = K(2.0) * M0[1] * M0[2] * dL_da + (M0[1] * M1[2] + M0[2] * M1[1]) * dL_db + K(2.0) * M1[1] * M1[2] * dL_dc;

    REAL v0[3], v1[3]; /* Sigma * M0, Sigma * M1 */
    v0[0] = c6[0] * M0[0] + c6[1] * M0[1] + c6[2] * M0[2];
    v0[1] = c6[1] * M0[0] + c6[3] * M0[1] + c6[4] * M0[2];
    v0[2] = c6[2] * M0[0] + c6[4] * M0[1] + c6[5] * M0[2];
    v1[0] = c6[0] * M1[0] + c6[1] * M1[1] + c6[2] * M1[2];
    v1[1] = c6[1] * M1[0] + c6[3] * M1[1] + c6[4] * M1[2];
    v1[2] = c6[2] * M1[0] + c6[4] * M1[1] + c6[5] * M1[2];
    REAL dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
    for (int j = 0; j < 3; j++) {
      const REAL dM0 = K(2.0) * v0[j] * dL_da + v1[j] * dL_db;
      const REAL dM1 = K(2.0) * v1[j] * dL_dc + v0[j] * dL_db;
      const REAL W0 = V[4 * j + 0], W1 = V[4 * j + 1], W2 = V[4 * j + 2];
      dJ00 += W0 * dM0; dJ02 += W2 * dM0; dJ11 += W1 * dM1; dJ12 += W2 * dM1;
    }
    const REAL tz = K(1.0) / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    const REAL dtx = e.xmul * -fx * tz2 * dJ02;
    const REAL dty = e.ymul * -fy * tz2 * dJ12;
    const REAL dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (K(2.0) * fx * e.tx) * tz3 * dJ02 +
                     (K(2.0) * fy * e.ty) * tz3 * dJ12;
    for (int j = 0; j < 3; j++)
      dmean[j] += V[4 * j + 0] * dtx + V[4 * j + 1] * dty + V[4 * j + 2] * dtz;

    /* (b) screen-space mean -> 3D mean through the perspective divide */
    REAL ph[4];
    xform44(PV, p, ph);
    const REAL mw = K(1.0) / (ph[3] + K(GSR_W_EPS));
    const REAL mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
    const REAL g2x = dL_dmean2D[2 * i], g2y = dL_dmean2D[2 * i + 1];
    for (int j = 0; j < 3; j++)
      dmean[j] += (PV[4 * j + 0] * mw - PV[4 * j + 3] * mul1) * g2x +
                  (PV[4 * j + 1] * mw - PV[4 * j + 3] * mul2) * g2y;

    /* (c) colour -> SH coefficients and view direction */
    if (shs && dL_dshs) {
      REAL dir[3] = {p[0] - cam->campos[0], p[1] - cam->campos[1], p[2] - cam->campos[2]};
      const REAL len = r_sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
      const REAL x = dir[0] / len, y = dir[1] / len, z = dir[2] / len;
      REAL bas[16];
      const int D = cam->D, n = ncoef(D);
      sh_basis(D, x, y, z, bas);
      const REAL* sh = shs + (size_t)i * cam->M * 3;
      REAL* dsh = dL_dshs + (size_t)i * cam->M * 3;
      REAL dRGB[3];
      for (int ch = 0; ch < 3; ch++) dRGB[ch] = clamped[3 * i + ch] ? K(0.0) : dL_dcolor[3 * i + ch];
      for (int k = 0; k < n; k++)
        for (int ch = 0; ch < 3; ch++) dsh[3 * k + ch] = bas[k] * dRGB[ch];
      /* d basis / d (x,y,z) */
      REAL bx[16] = {0}, by[16] = {0}, bz[16] = {0};
      if (D > 0) {
        by[1] = -K(GSR_SH_C1); bz[2] = K(GSR_SH_C1); bx[3] = -K(GSR_SH_C1);
        if (D > 1) {
          const REAL xx = x * x, yy = y * y, zz = z * z;
          bx[4] = K(GSR_SH_C2_0) * y; by[4] = K(GSR_SH_C2_0) * x;
          by[5] = K(GSR_SH_C2_1) * z; bz[5] = K(GSR_SH_C2_1) * y;
          bx[6] = K(GSR_SH_C2_2) * K(-2.0) * x; by[6] = K(GSR_SH_C2_2) * K(-2.0) * y; bz[6] = K(GSR_SH_C2_2) * K(4.0) * z;
          bx[7] = K(GSR_SH_C2_3) * z; bz[7] = K(GSR_SH_C2_3) * x;
          bx[8] = K(GSR_SH_C2_4) * K(2.0) * x; by[8] = K(GSR_SH_C2_4) * K(-2.0) * y;
          if (D > 2) {
            bx[9] = K(GSR_SH_C3_0) * K(6.0) * x * y; by[9] = K(GSR_SH_C3_0) * (K(3.0) * xx - K(3.0) * yy);
            bx[10] = K(GSR_SH_C3_1) * y * z; by[10] = K(GSR_SH_C3_1) * x * z; bz[10] = K(GSR_SH_C3_1) * x * y;
            bx[11] = K(GSR_SH_C3_2) * K(-2.0) * x * y; by[11] = K(GSR_SH_C3_2) * (K(4.0) * zz - xx - K(3.0) * yy);
            bz[11] = K(GSR_SH_C3_2) * K(8.0) * y * z;
            bx[12] = K(GSR_SH_C3_3) * K(-6.0) * x * z; by[12] = K(GSR_SH_C3_3) * K(-6.0) * y * z;
            bz[12] = K(GSR_SH_C3_3) * (K(6.0) * zz - K(3.0) * xx - K(3.0) * yy);
            bx[13] = K(GSR_SH_C3_4) * (K(4.0) * zz - K(3.0) * xx - yy); by[13] = K(GSR_SH_C3_4) * K(-2.0) * x * y;
            bz[13] = K(GSR_SH_C3_4) * K(8.0) * x * z;
            bx[14] = K(GSR_SH_C3_5) * K(2.0) * x * z; by[14] = K(GSR_SH_C3_5) * K(-2.0) * y * z;
            bz[14] = K(GSR_SH_C3_5) * (xx - yy);
            bx[15] = K(GSR_SH_C3_6) * (K(3.0) * xx - K(3.0) * yy); by[15] = K(GSR_SH_C3_6) * K(-6.0) * x * y;
          }
        }
      }
      REAL ddir[3] = {0, 0, 0};
      for (int k = 1; k < n; k++) {
        const REAL s = sh[3 * k] * dRGB[0] + sh[3 * k + 1] * dRGB[1] + sh[3 * k + 2] * dRGB[2];
        ddir[0] += bx[k] * s; ddir[1] += by[k] * s; ddir[2] += bz[k] * s;
      }
      /* through n = v/|v|: dv = (dn - n (n.dn)) / |v| */
      const REAL nd = x * ddir[0] + y * ddir[1] + z * ddir[2];
      dmean[0] += (ddir[0] - x * nd) / len;
      dmean[1] += (ddir[1] - y * nd) / len;
      dmean[2] += (ddir[2] - z * nd) / len;
    }
    dL_dmeans3D[3 * i] = dmean[0]; dL_dmeans3D[3 * i + 1] = dmean[1]; dL_dmeans3D[3 * i + 2] = dmean[2];

    /* (d) cov3D -> scale, rotation */
    if (scales && dL_dscales && dL_drotations) {
      const REAL* q = rotations + 4 * i;
      const REAL mod = cam->scale_modifier;
      REAL R[9];
      quat_to_rot(q, R);
      const REAL S[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
      REAL Gm[9] = {dc6[0], K(0.5) * dc6[1], K(0.5) * dc6[2],
                    K(0.5) * dc6[1], dc6[3], K(0.5) * dc6[4],
                    K(0.5) * dc6[2], K(0.5) * dc6[4], dc6[5]};
      REAL dLm[9]; /* dL/dL = 2 G L,  L = R diag(S) */
      for (int r = 0; r < 3; r++)
        for (int j = 0; j < 3; j++) {
          REAL s = 0;
          for (int k = 0; k < 3; k++) s += Gm[3 * r + k] * (R[3 * k + j] * S[j]);
          dLm[3 * r + j] = K(2.0) * s;
        }
      REAL dR[9];
      for (int j = 0; j < 3; j++) {
        REAL ds = 0;
        for (int r = 0; r < 3; r++) { ds += R[3 * r + j] * dLm[3 * r + j]; dR[3 * r + j] = S[j] * dLm[3 * r + j]; }
        dL_dscales[3 * i + j] = mod * ds;
      }
      const REAL r = q[0], x = q[1], y = q[2], z = q[3];
      dL_drotations[4 * i + 0] = K(2.0) * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
      dL_drotations[4 * i + 1] = K(2.0) * (y * dR[1] + z * dR[2] + y * dR[3] - K(2.0) * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - K(2.0) * x * dR[8]);
      dL_drotations[4 * i + 2] = K(2.0) * (K(-2.0) * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - K(2.0) * y * dR[8]);
      dL_drotations[4 * i + 3] = K(2.0) * (K(-2.0) * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - K(2.0) * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
  }
}

/* ------------------------------------------------------------------------ */
/* Stage fragments exported on their own so that the golden vectors made     */
/* from the reference's Python can pin them (same inline code as above).     */
/* ------------------------------------------------------------------------ */

/* rgb_raw[n][ch] = sum_k basis_k(dir_n) * sh[n][k][ch]  (no +0.5, no clamp):
 * eval_sh of utils/sh_utils.py:57-112 on (N,M,3) coefficient-major input. */
void gso_sh_eval(int D, int M, int N, const REAL* dirs, const REAL* shs, REAL* rgb_raw) {
  for (int i = 0; i < N; i++) {
    REAL b[16];
    sh_basis(D, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], b);
    const REAL* sh = shs + (size_t)i * M * 3;
    for (int ch = 0; ch < 3; ch++) {
      REAL acc = b[0] * sh[ch];
      for (int k = 1; k < ncoef(D); k++) acc = acc + b[k] * sh[3 * k + ch];
      rgb_raw[3 * i + ch] = acc;
    }
  }
}

/* build_covariance_from_scaling_rotation (scene/gaussian_model.py:27-31). */
void gso_cov3d(int N, const REAL* scales, REAL mod, const REAL* rots, REAL* cov6) {
  for (int i = 0; i < N; i++) cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov6 + 6 * i);
}
