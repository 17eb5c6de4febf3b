"""CPU restatement of distCUDA2 (the reference's simple_knn extension; call site
scene/gaussian_model.py:134): mean of the three smallest squared distances from every point to
the other points.  Test infrastructure only.

Parity unpinned: the extension's source is an un-vendored submodule (.gitmodules:1-3, directory
empty) and the reference holds no fixture of its output; the quantity is restated from its use
and from the published module (exact 3-nearest-neighbour search).
"""
import numpy as np


def _dist2(a, b):
    """((dx*dx + dy*dy) + dz*dz) in fp32, rounded after every operation; a (n,1,3), b (1,m,3)."""
    d = (a - b).astype(np.float32)
    return ((d[..., 0] * d[..., 0]).astype(np.float32) + (d[..., 1] * d[..., 1]).astype(np.float32)).astype(np.float32) \
        + (d[..., 2] * d[..., 2]).astype(np.float32)


def mean_dist2_bruteforce(points, chunk=256):
    """All pairs, fp32, the same operation sequence as the GPU kernel: comparable bit for bit."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    P = pts.shape[0]
    out = np.zeros(P, np.float32)
    k = min(3, P - 1)
    if k <= 0:
        return out
    for s in range(0, P, chunk):
        e = min(P, s + chunk)
        d = _dist2(pts[s:e, None, :], pts[None, :, :])
        d[np.arange(e - s), np.arange(s, e)] = np.inf
        best = np.sort(np.partition(d, k - 1, axis=1)[:, :k], axis=1)
        acc = best[:, 0].copy()
        for j in range(1, k):
            acc = (acc + best[:, j]).astype(np.float32)
        out[s:e] = acc / np.float32(k)
    return out


def mean_dist2_kdtree(points):
    """Large clouds: neighbours from a k-d tree (float64 search), distances recomputed in fp32."""
    from scipy.spatial import cKDTree
    pts = np.ascontiguousarray(points, dtype=np.float32)
    _, idx = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    nb = pts[idx[:, 1:4]]
    d = _dist2(pts[:, None, :], nb)
    d = np.sort(d, axis=1)
    return (((d[:, 0] + d[:, 1]).astype(np.float32) + d[:, 2]).astype(np.float32) / np.float32(3)).astype(np.float32)
