"""The two CPU restatements of distCUDA2 agree with each other and with a direct definition."""
import numpy as np


def test_bruteforce_matches_definition_and_kdtree():
    from oracle import knn_ref
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(400, 3)).astype(np.float32)
    want = np.empty(400, np.float64)
    for i in range(400):
        d = ((pts.astype(np.float64) - pts[i].astype(np.float64)) ** 2).sum(1)
        d[i] = np.inf
        want[i] = np.sort(d)[:3].mean()
    a, b = knn_ref.mean_dist2_bruteforce(pts), knn_ref.mean_dist2_kdtree(pts)
    assert np.allclose(a, want, rtol=1e-5) and np.allclose(b, want, rtol=1e-5)
    assert knn_ref.mean_dist2_bruteforce(pts[:1])[0] == 0.0
    two = knn_ref.mean_dist2_bruteforce(pts[:2])
    assert two[0] == two[1] > 0
