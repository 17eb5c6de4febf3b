import sys, os, math, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gaussian-splatting-own_amd"))
import diff_gaussian_rasterization as dgr
from gsplat_host.cameras import orbit_camera
from gsplat_host.model import GaussianParams
from gsplat_host.synthetic import make_cloud
dev = torch.device("cuda")
cloud = make_cloud(1_000_000, sh_degree=3, seed=0)
model = GaussianParams(cloud, device=dev)
cam = orbit_camera(1920, 1080, device=dev)
bg = torch.zeros(3, device=dev)
s = dgr.GaussianRasterizationSettings(image_height=1080, image_width=1920, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center, prefiltered=False, debug=False)
with torch.no_grad():
    st = dgr.rasterize_forward(s, model.get_xyz, model.get_opacity, shs=model.get_features, scales=model.get_scaling, rotations=model.get_rotation)
torch.cuda.synchronize()
H, W = 1080, 1920
nc = st.n_contrib.to(torch.int64)
pad = torch.zeros((1088, 1920), dtype=torch.int64, device=dev); pad[:1080] = nc
tl = pad.reshape(68, 16, 120, 16).amax(dim=(1, 3)).float()          # per-tile deepest contributor
ln = (st.ranges[:, 1] - st.ranges[:, 0]).reshape(68, 120).float()
print("tile list length: mean %.0f max %.0f" % (ln.mean(), ln.max()))
print("tile_last: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (tl.mean(), tl.median(), tl.flatten().kthvalue(int(0.9*8160)).values, tl.flatten().kthvalue(int(0.99*8160)).values, tl.max()))
q = pad.reshape(68, 2, 8, 120, 2, 8).amax(dim=(2, 5)).float()        # per quadrant
print("quadrant_last mean %.0f ; mean of tile max %.0f ; ratio %.2f" % (q.mean(), tl.mean(), tl.mean() / q.mean()))
print("rows (tile_last mean per tile row):", [int(x) for x in tl.mean(dim=1)[::4]])
print("sum tile_last %.0f ; top 5%% tiles share %.2f" % (tl.sum(), tl.flatten().sort(descending=True).values[:408].sum() / tl.sum()))
