"""Shared by the experiment scripts: the depth keys and sort values of a benchmark scene's view."""
import math

import torch


def scene_state(workload, dev):
    import bench
    import diff_gaussian_rasterization as dgr
    sc = bench.Scene(workload, dev, 0)
    m, cam = sc.model, sc.cam
    s = dgr.GaussianRasterizationSettings(
        image_height=sc.H, image_width=sc.W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=sc.bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=sc.deg, campos=cam.camera_center, prefiltered=False, debug=False)
    lib, ptr = dgr._lib, dgr._ptr
    P, W, H = sc.P, sc.W, sc.H
    u32 = dict(dtype=torch.int32, device=dev)
    totals = torch.empty((lib.gsr_totals_words(P),), **u32)
    rec, radii, keys = torch.empty((P, 16), **u32), torch.empty((P,), **u32), torch.empty((P,), **u32)
    sv = torch.empty((P,), dtype=torch.int64, device=dev)
    with torch.no_grad():
        rc = lib.gsr_preprocess_fwd_raw(P, sc.deg, 15, ptr(m._xyz), ptr(m._scaling), 1.0, ptr(m._rotation), ptr(m._opacity),
                                        ptr(m._features_dc), ptr(m._features_rest), ptr(cam.world_view_transform),
                                        ptr(cam.full_proj_transform), ptr(cam.camera_center), W, H, float(s.tanfovx),
                                        float(s.tanfovy), 1, ptr(rec), ptr(radii), None, ptr(keys), None, ptr(sv),
                                        ptr(totals), None, None, dgr._stream())
    assert rc == 0, lib.gsr_last_error()
    torch.cuda.synchronize()
    return P, keys, sv
