"""The model file format (scene/gaussian_model.py:177-256): property order, channel-major SH
layout on disk, and a write/read round trip."""
import os

import numpy as np
import torch

from conftest import ROOT  # noqa: F401  (puts the package on sys.path)


def test_attribute_order_matches_the_reference_list():
    from gsplat_host.ply import construct_list_of_attributes
    names = construct_list_of_attributes()
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"]
    assert names[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9] == "f_rest_0" and names[53] == "f_rest_44" and names[54] == "opacity"
    assert names[55:58] == ["scale_0", "scale_1", "scale_2"] and names[58:] == ["rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(names) == 62


def test_round_trip_and_on_disk_layout(tmp_path):
    from gsplat_host.ply import load_ply, read_vertex_table, save_ply
    g = torch.Generator().manual_seed(0)
    P = 37
    m = {"xyz": torch.randn(P, 3, generator=g), "features_dc": torch.randn(P, 1, 3, generator=g),
         "features_rest": torch.randn(P, 15, 3, generator=g), "opacity": torch.randn(P, 1, generator=g),
         "scaling": torch.randn(P, 3, generator=g), "rotation": torch.randn(P, 4, generator=g)}
    path = os.path.join(tmp_path, "point_cloud", "iteration_7", "point_cloud.ply")
    save_ply(path, m["xyz"], m["features_dc"], m["features_rest"], m["opacity"], m["scaling"], m["rotation"])
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"end_header\n") + len(b"end_header\n")].decode("ascii").split("\n")
    assert head[0] == "ply" and head[1] == "format binary_little_endian 1.0" and head[2] == f"element vertex {P}"
    assert head[3] == "property float x" and head[3 + 61] == "property float rot_3" and head[3 + 62] == "end_header"
    assert len(raw) == len("\n".join(head)) + P * 62 * 4
    col = read_vertex_table(path)
    # channel-major SH on disk: f_rest_k holds coefficient k % 15 of channel k // 15
    assert np.allclose(col["f_rest_16"], m["features_rest"][:, 1, 1].numpy())
    assert np.allclose(col["f_dc_2"], m["features_dc"][:, 0, 2].numpy())
    assert np.all(col["nx"] == 0)
    back = load_ply(path, max_sh_degree=3, device="cpu")
    for k in m:
        assert back[k].shape == m[k].shape and torch.equal(back[k], m[k]), k
