"""gsr_bin_sort_static, the entry point that allocates nothing (its scratch is a device array inside the
library), against gsr_bin_sort through the C ABI: the same sorted values bit for bit, on two calls in a
row of which the second is smaller (the array keeps the first call's contents), with the free device
memory unchanged across the calls; sizes beyond the array's capacity are refused with a clear error."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CULLED = 0xFFFFFFFF


def _inputs(P, seed):
    rng = np.random.default_rng(seed)
    z = rng.uniform(0.2, 30.0, P).astype(np.float32)
    keys = z.view(np.uint32).copy()
    keys[rng.random(P) < 0.25] = CULLED
    vals = rng.integers(0, 2**63, P, dtype=np.uint64)
    return keys, vals


def _existing(dgr, k, v):
    P = k.numel()
    out = torch.full((P,), -0x1111111111111112, dtype=torch.int64, device="cuda")
    nb = dgr._lib.gsr_bin_sort_bytes(P)
    scratch = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    dgr._run("bin_sort", dgr._lib.gsr_bin_sort, P, dgr._ptr(k), dgr._ptr(v), dgr._ptr(out), dgr._ptr(scratch), nb,
             dgr._stream())
    assert dgr._lib.gsr_bin_sort_status(P, dgr._ptr(scratch), nb, dgr._stream()) == 0
    return out


def test_static_scratch_sort_matches_the_allocating_one_twice_in_a_row():
    import diff_gaussian_rasterization as dgr
    lib = dgr._lib
    lib.gsr_bin_sort_static.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
    cases = []
    for P, seed in ((1_000_000, 1), (300_001, 2)):            # the second call is smaller than the first
        keys, vals = _inputs(P, seed)
        k = torch.from_numpy(keys.view(np.int32)).cuda()
        v = torch.from_numpy(vals.view(np.int64)).cuda()
        want = _existing(dgr, k, v)
        got = torch.full((P,), -0x1111111111111112, dtype=torch.int64, device="cuda")
        cases.append((P, keys, vals, k, v, want, got))
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    words = []
    for P, keys, vals, k, v, want, got in cases:
        word = C.c_void_p()
        rc = lib.gsr_bin_sort_static(P, dgr._ptr(k), dgr._ptr(v), dgr._ptr(got), C.byref(word), dgr._stream())
        assert rc == 0, lib.gsr_last_error().decode()
        words.append(word.value)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before           # nothing was allocated or freed
    assert words[0] and words[0] == words[1] and words[0] % 4 == 0   # one array per GPU, the same on every call
    for P, keys, vals, k, v, want, got in cases:
        order = np.argsort(keys, kind="stable")
        order = order[keys[order] != CULLED]
        V = len(order)
        g = got.cpu().numpy().view(np.uint64)
        assert np.array_equal(g, want.cpu().numpy().view(np.uint64))          # the existing entry point, bit for bit
        assert np.array_equal(g[:V], vals[order]) and (g[V:] == np.uint64(0xEEEEEEEEEEEEEEEE)).all()
    # the sort's error word, read back through the address the call returned, says the last sort was healthy
    try:
        hip = C.CDLL("libamdhip64.so")
    except OSError:
        hip = None                      # no unversioned runtime library on the loader path: skip this read-back
    if hip is not None:
        err = (C.c_uint32 * 1)()
        assert hip.hipMemcpy(err, C.c_void_p(words[1]), 4, 2) == 0 and err[0] == 0   # 2 = device to host

    # capacity: one key more than the array holds is refused before anything is launched
    rc = lib.gsr_bin_sort_static((1 << 20) + 1, dgr._ptr(cases[0][3]), dgr._ptr(cases[0][4]), dgr._ptr(cases[0][6]), None,
                                 dgr._stream())
    assert rc == -5 and "exceed the static scratch's capacity" in lib.gsr_last_error().decode()
    assert lib.gsr_bin_sort_static(0, None, None, None, None, None) == 0
