/*
 * Numeric constants of the tile-based Gaussian rasterizer, in one place.
 *
 * The rasterizer the reference imports (gaussian_renderer/__init__.py:14,
 * package diff_gaussian_rasterization) is an un-vendored, un-pinned git
 * submodule (.gitmodules:4-6) whose directory is empty in the reference tree.
 * Every value tagged [UPSTREAM] below is therefore the published behaviour of
 * that third-party module (3D Gaussian Splatting, Kerbl et al. 2023, and the
 * public graphdeco-inria/diff-gaussian-rasterization project of mid-2023),
 * restated here; none of them can be cited to a file:line of the reference.
 * Values tagged [REF file:line] are pinned by the reference's own Python.
 *
 * Plain C, usable from HIP device code, host C++ and the C oracle alike.
 */
#ifndef GSR_CONSTANTS_H
#define GSR_CONSTANTS_H

/* Screen is binned in square tiles of this many pixels. [UPSTREAM] */
#define GSR_TILE_X 16
#define GSR_TILE_Y 16
#define GSR_TILE_PIXELS (GSR_TILE_X * GSR_TILE_Y)
/* Side, in tiles, of the supercells (and width of the cells) of the binning's range partitions. */
#define GSR_SUPERCELL 16

/* A Gaussian whose view-space depth is <= this is culled. [UPSTREAM] */
#define GSR_NEAR_CULL 0.2f
/* Added to the homogeneous w before the perspective divide.
 * [REF utils/graphics_utils.py:28 uses the same 1e-7 in geom_transform_points] */
#define GSR_W_EPS 0.0000001f
/* x/z and y/z are clamped to +-(this * tan(fov/2)) before the EWA Jacobian. [UPSTREAM] */
#define GSR_FRUSTUM_CLAMP 1.3f
/* Low-pass dilation added to the diagonal of the 2D covariance, in px^2. [UPSTREAM] */
#define GSR_COV2D_DILATION 0.3f
/* Floor under the discriminant used for the eigenvalues of the 2D covariance. [UPSTREAM] */
#define GSR_LAMBDA_FLOOR 0.1f
/* Screen radius = ceil(this * sqrt(largest eigenvalue)). [UPSTREAM] */
#define GSR_RADIUS_SIGMAS 3.0f

/* Blending. [UPSTREAM] */
#define GSR_ALPHA_MAX 0.99f            /* alpha = min(this, opacity * G)          */
#define GSR_ALPHA_MIN (1.0f / 255.0f)  /* contributions below this are skipped    */
#define GSR_T_STOP 0.0001f             /* a pixel is finished once T would drop below */

/* Backward of conic = inverse(cov2D): 1 / (det^2 + this). [UPSTREAM] */
#define GSR_DET2_EPS 0.0000001f

/* Colour = SH(dir) + this, then clamped at >= 0.
 * [REF gaussian_renderer/__init__.py:78] */
#define GSR_SH_OFFSET 0.5f

/* Real spherical-harmonics constants. [REF utils/sh_utils.py:26-43] */
#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f
#define GSR_SH_C2_0 1.0925484305920792f
#define GSR_SH_C2_1 -1.0925484305920792f
#define GSR_SH_C2_2 0.31539156525252005f
#define GSR_SH_C2_3 -1.0925484305920792f
#define GSR_SH_C2_4 0.5462742152960396f
#define GSR_SH_C3_0 -0.5900435899266435f
#define GSR_SH_C3_1 2.890611442640554f
#define GSR_SH_C3_2 -0.4570457994644658f
#define GSR_SH_C3_3 0.3731763325901154f
#define GSR_SH_C3_4 -0.4570457994644658f
#define GSR_SH_C3_5 1.445305721320277f
#define GSR_SH_C3_6 -0.5900435899266435f

/* Highest SH degree the kernels evaluate (16 coefficients per channel).
 * [REF arguments/__init__.py:49 sh_degree = 3] */
#define GSR_MAX_SH_DEGREE 3

#endif /* GSR_CONSTANTS_H */
