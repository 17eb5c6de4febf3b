"""gsr_bin_sort through the C ABI against numpy's stable argsort: culled keys dropped, ties in
input order, sizes around the workgroup and look-back group boundaries, both items-per-thread
variants."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CULLED = 0xFFFFFFFF


def _sort(keys: np.ndarray, vals: np.ndarray, want_status: bool = False):
    """The sorted values; the status call must report a healthy sort unless the caller wants the code itself."""
    import diff_gaussian_rasterization as dgr
    P = len(keys)
    k = torch.from_numpy(keys.view(np.int32)).cuda()
    v = torch.from_numpy(vals.view(np.int64)).cuda()
    out = torch.full((P,), -0x1111111111111112, dtype=torch.int64, device="cuda")   # 0xEEEE...EE
    nb = dgr._lib.gsr_bin_sort_bytes(P)
    scratch = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    dgr._run("bin_sort", dgr._lib.gsr_bin_sort, P, dgr._ptr(k), dgr._ptr(v), dgr._ptr(out), dgr._ptr(scratch), nb,
             dgr._stream())
    rc = dgr._lib.gsr_bin_sort_status(P, dgr._ptr(scratch), nb, dgr._stream())     # waits for the stream
    got = out.cpu().numpy().view(np.uint64)
    if want_status:
        return got, rc, dgr._lib.gsr_last_error().decode()
    assert rc == 0, dgr._lib.gsr_last_error().decode()
    return got


def _check(keys, vals):
    got = _sort(keys, vals)
    order = np.argsort(keys, kind="stable")
    order = order[keys[order] != CULLED]
    V = len(order)
    assert np.array_equal(got[:V], vals[order])
    assert (got[V:] == np.uint64(0xEEEEEEEEEEEEEEEE)).all()      # nothing written behind the visible ones


def _depth_keys(rng, P, culled_fraction, lo=0.2, hi=30.0):
    z = rng.uniform(lo, hi, P).astype(np.float32)
    keys = z.view(np.uint32).copy()
    keys[rng.random(P) < culled_fraction] = CULLED
    return keys


@pytest.mark.parametrize("P", [1, 63, 64, 65, 2047, 2048, 2049, 4095, 4096, 4097, 65536, 65537, 199_999, 200_000, 300_001])
def test_sizes_around_the_block_boundaries(P):
    rng = np.random.default_rng(P)
    _check(_depth_keys(rng, P, 0.2), rng.integers(0, 2**63, P, dtype=np.uint64))


def test_one_million_keys_and_the_eight_item_variant():
    rng = np.random.default_rng(7)
    for P in (1_000_000, 5_000_000):
        _check(_depth_keys(rng, P, 0.3), rng.integers(0, 2**63, P, dtype=np.uint64))


def test_exclusive_device_order_gives_the_same_result():
    """With the caller's word that it has the GPU to itself (this test process does), a sort whose whole grid is
    resident gives workgroup b chunk b instead of a ticket; the result is the same.  The sizes cover all three
    keys-per-thread variants, one grid that is resident at once and one (5M keys) that is not and keeps its tickets."""
    import diff_gaussian_rasterization as dgr
    rng = np.random.default_rng(11)
    dgr.assume_exclusive_device(True)
    try:
        for P in (4097, 300_001, 1_000_000, 5_000_000):
            _check(_depth_keys(rng, P, 0.25), rng.integers(0, 2**63, P, dtype=np.uint64))
    finally:
        dgr.assume_exclusive_device(False)


def test_ties_keep_input_order():
    rng = np.random.default_rng(3)
    P = 200_000
    keys = rng.choice(np.array([0x3F800000, 0x3F800001, 0x40000000, 0x40490FDB, CULLED], dtype=np.uint32), P)
    _check(keys, np.arange(P, dtype=np.uint64))


def test_all_culled_none_culled_and_arbitrary_bit_patterns():
    rng = np.random.default_rng(5)
    P = 70_000
    vals = rng.integers(0, 2**63, P, dtype=np.uint64)
    _check(np.full(P, CULLED, dtype=np.uint32), vals)
    _check(_depth_keys(rng, P, 0.0), vals)
    keys = rng.integers(0, 2**32 - 1, P, dtype=np.uint64).astype(np.uint32)     # every byte of the key varies
    _check(keys, vals)
    _check(np.full(P, 0x40A00000, dtype=np.uint32), vals)                       # one key: one digit per pass


def test_narrow_depth_range_many_groups():
    """All keys inside one power of two (top bytes constant) and more than 16 workgroups, so that
    the group level of the look-back is exercised with every count in one bin."""
    rng = np.random.default_rng(11)
    P = 16 * 4096 * 3 + 17
    _check(_depth_keys(rng, P, 0.1, lo=5.0, hi=5.5), rng.integers(0, 2**63, P, dtype=np.uint64))


@pytest.mark.parametrize("culled_fraction", [0.1, 0.0])
def test_a_sort_that_runs_out_of_patience_says_so_and_stays_in_its_buffers(culled_fraction):
    """With a patience of one look-back step a workgroup that finds its predecessors' counts missing twice
    gives up: every kernel still ends, and the status call reports the failure (or, if no workgroup
    happened to wait, a complete sort).  A failed sort must not write outside its scratch and the P
    output words, whatever the scratch held before: the scratch is poisoned with 0xFF (keys that would
    rank last, counts that are huge), input, output and scratch sit in one allocation between canaries,
    and with nothing culled every slot of every buffer is in use.  The default patience is back afterwards."""
    import diff_gaussian_rasterization as dgr
    rng = np.random.default_rng(13)
    P = 1_000_000
    keys, vals = _depth_keys(rng, P, culled_fraction), rng.integers(0, 2**63, P, dtype=np.uint64)
    order = np.argsort(keys, kind="stable")
    order = order[keys[order] != CULLED]
    nb = dgr._lib.gsr_bin_sort_bytes(P)
    pad = 1 << 16
    # [canary | out (8P) | canary | scratch (nb) | canary], everything 256-byte aligned
    out_off, scr_off = pad, pad + 8 * P + pad
    scr_off = (scr_off + 255) // 256 * 256
    arena = torch.full((scr_off + nb + pad,), 0xA5, dtype=torch.uint8, device="cuda")
    base = arena.data_ptr()
    assert base % 256 == 0
    arena[scr_off:scr_off + nb] = 0xFF
    arena[out_off:out_off + 8 * P] = 0xEE
    k = torch.from_numpy(keys.view(np.int32)).cuda()
    v = torch.from_numpy(vals.view(np.int64)).cuda()
    import ctypes as C
    dgr._lib.gsr_debug_sort_spin_limit(1)
    try:
        dgr._run("bin_sort", dgr._lib.gsr_bin_sort, P, dgr._ptr(k), dgr._ptr(v), C.c_void_p(base + out_off),
                 C.c_void_p(base + scr_off), nb, dgr._stream())
        rc = dgr._lib.gsr_bin_sort_status(P, C.c_void_p(base + scr_off), nb, dgr._stream())
        message = dgr._lib.gsr_last_error().decode()
    finally:
        dgr._lib.gsr_debug_sort_spin_limit(0)
    host = arena.cpu().numpy()
    for lo, hi in ((0, out_off), (out_off + 8 * P, scr_off), (scr_off + nb, len(host))):
        assert (host[lo:hi] == 0xA5).all(), "the sort wrote outside its buffers"
    assert np.array_equal(k.cpu().numpy().view(np.uint32), keys) and np.array_equal(v.cpu().numpy().view(np.uint64), vals)
    got = host[out_off:out_off + 8 * P].view(np.uint64)
    print("impatient sort: rc", rc, flush=True)
    if rc == 0:
        assert np.array_equal(got[:len(order)], vals[order])
    else:
        assert rc == -4 and "gave up" in message
        # what was written is made of input values (or was never written)
        written = got[got != np.uint64(0xEEEEEEEEEEEEEEEE)]
        assert np.isin(written, vals).all()
    _check(keys, vals)
