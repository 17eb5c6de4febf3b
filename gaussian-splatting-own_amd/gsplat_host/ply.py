"""The reference's on-disk model format: a binary little-endian PLY with one float32 property per
scalar of every Gaussian, in the order of ``construct_list_of_attributes``
(``scene/gaussian_model.py:177-189``), written by ``save_ply`` (``:191-208``) and read by
``load_ply`` (``:215-256``).  The reference goes through the ``plyfile`` package, which this image
lacks; the format itself is plain PLY 1.0 and is produced / parsed here directly with numpy.

Parity unpinned at the byte level (no PLY fixture in the reference tree, no plyfile to compare
with): the header follows the PLY specification and the property names, order and per-tensor
layouts follow the reference's code.
"""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np
import torch


def construct_list_of_attributes(n_dc: int = 3, n_rest: int = 45, n_scale: int = 3, n_rot: int = 4) -> List[str]:
    """scene/gaussian_model.py:177-189."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation) -> None:
    """Raw (pre-activation) parameters, as the model stores them: xyz (P,3), features_dc (P,1,3),
    features_rest (P,K,3), opacity (P,1), scaling (P,3), rotation (P,4).  SH coefficients go out
    channel-major, as the reference's transpose(1, 2).flatten(start_dim=1) makes them (:196-197)."""
    t = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    xyz, opacity, scaling, rotation = t(xyz), t(opacity), t(scaling), t(rotation)
    f_dc = np.ascontiguousarray(np.transpose(t(features_dc), (0, 2, 1))).reshape(xyz.shape[0], -1)
    f_rest = np.ascontiguousarray(np.transpose(t(features_rest), (0, 2, 1))).reshape(xyz.shape[0], -1)
    normals = np.zeros_like(xyz)
    table = np.concatenate((xyz, normals, f_dc, f_rest, opacity, scaling, rotation), axis=1).astype("<f4")
    names = construct_list_of_attributes(f_dc.shape[1], f_rest.shape[1], scaling.shape[1], rotation.shape[1])
    assert table.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {table.shape[0]}"]
    header += [f"property float {n}" for n in names]
    header.append("end_header")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())


def read_vertex_table(path: str) -> Dict[str, np.ndarray]:
    """Property name -> (P,) float64 column of the file's vertex element.  Accepts what save_ply
    and plyfile write: binary little-endian, one vertex element first, scalar properties."""
    sizes = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
             "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
             "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            w = line.decode("ascii").split()
            if not w or w[0] == "comment":
                continue
            if w[0] == "format":
                fmt = w[1]
            elif w[0] == "element":
                in_vertex = w[1] == "vertex" and count is None
                if in_vertex:
                    count = int(w[2])
            elif w[0] == "property" and in_vertex:
                if w[1] == "list":
                    raise ValueError(f"{path}: list property in the vertex element")
                props.append((w[2], sizes[w[1]]))
            elif w[0] == "end_header":
                break
        if fmt != "binary_little_endian" or count is None:
            raise ValueError(f"{path}: expected a binary_little_endian PLY with a vertex element, got {fmt}")
        rec = np.dtype(props)
        data = np.frombuffer(f.read(rec.itemsize * count), dtype=rec, count=count)
    return {n: data[n].astype(np.float64) for n, _ in props}


def load_ply(path: str, max_sh_degree: int = 3, device="cuda") -> Dict[str, torch.Tensor]:
    """The six raw parameter tensors in the model's layouts (scene/gaussian_model.py:215-254):
    f_rest_* columns sorted by their numeric suffix, reshaped (P,3,K) and transposed to (P,K,3)."""
    col = read_vertex_table(path)
    xyz = np.stack((col["x"], col["y"], col["z"]), axis=1)
    opacity = col["opacity"][:, None]
    f_dc = np.stack((col["f_dc_0"], col["f_dc_1"], col["f_dc_2"]), axis=1)[:, :, None]            # (P,3,1)
    by_suffix = lambda prefix: sorted((n for n in col if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    rest = by_suffix("f_rest_")
    if len(rest) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest)} f_rest properties, expected {3 * (max_sh_degree + 1) ** 2 - 3}")
    f_rest = np.stack([col[n] for n in rest], axis=1).reshape(xyz.shape[0], 3, (max_sh_degree + 1) ** 2 - 1)
    scaling = np.stack([col[n] for n in by_suffix("scale_")], axis=1)
    rotation = np.stack([col[n] for n in by_suffix("rot")], axis=1)
    t = lambda a: torch.tensor(a, dtype=torch.float, device=device)
    return {"xyz": t(xyz), "features_dc": t(f_dc).transpose(1, 2).contiguous(),
            "features_rest": t(f_rest).transpose(1, 2).contiguous(), "opacity": t(opacity), "scaling": t(scaling),
            "rotation": t(rotation)}
