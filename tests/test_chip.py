"""CPU: the XCD-aware block order is a pure index map (include/agt_hip.h agt_xcd_tile_order = agt_kernels.h agt_xcd_order,
the function every kernel with an L2-aware order calls) -- checked for XCD counts 1 / 2 / 4 / 8 without a GPU; and the
chip description agt_create derives from the device properties."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from accurate_aprilgroup_tracking_amd import hiplib
    L = ctypes.CDLL(hiplib.LIB_PATH)
    L.agt_xcd_tile_order.argtypes = [ctypes.c_int] * 3
    L.agt_xcd_tile_order.restype = ctypes.c_int
    return L


@pytest.mark.parametrize("xcds", [1, 2, 4, 8])
def test_xcd_order_is_a_permutation_with_contiguous_runs_per_xcd(lib, xcds):
    for nblk in (xcds, 8 * xcds, 24 * xcds, 8 * 5 * 64, 4608 // 8 * xcds):
        if nblk % xcds:
            continue
        items = np.array([lib.agt_xcd_tile_order(b, nblk, xcds) for b in range(nblk)])
        assert sorted(items.tolist()) == list(range(nblk)), "not a permutation"
        per = nblk // xcds
        for k in range(xcds):
            mine = items[k::xcds]                  # the blocks the hardware deals to XCD k (round robin), in dispatch order
            assert np.array_equal(mine, np.arange(k * per, (k + 1) * per)), "XCD %d does not walk a contiguous run" % k
        if xcds == 1:
            assert np.array_equal(items, np.arange(nblk)), "one XCD: the plain order"


def test_xcd_order_rejects_bad_arguments(lib):
    assert lib.agt_xcd_tile_order(0, 12, 8) == -1         # grid not a multiple of the XCD count
    assert lib.agt_xcd_tile_order(0, 16, 3) == -1         # not a power of two
    assert lib.agt_xcd_tile_order(16, 16, 8) == -1        # block past the grid
    assert lib.agt_xcd_tile_order(-1, 16, 8) == -1
    assert lib.agt_xcd_tile_order(5, 16, 8) == 5 * 2 + 0


def test_eight_xcd_map_is_the_literal_of_rounds_1_to_4(lib):
    """the shipped orders of rounds 1-4 were (b & 7) * (n >> 3) + (b >> 3): unchanged on a whole MI355X"""
    n = 8 * 37
    for b in range(n):
        assert lib.agt_xcd_tile_order(b, n, 8) == (b & 7) * (n >> 3) + (b >> 3)


def test_lk_residency_cap_lds_request(lib):
    """agt_lk_lds_request (round 6): the LDS a one-wave LK workgroup asks for so that AT MOST n workgroups fit in a CU's 160 KB whatever the
    hardware's allocation granule (n + 1 never fit), up to 16 per CU exactly n for granules up to 512 B and n - 1 at worst for 1,024 / 1,280 B (exactly n
    for those too at the library's own counts, 8 and 10), monotone, and 0 where a cap cannot be expressed (none asked for; one per CU would need more than a workgroup's 64 KB).
    (The first form of the rule -- one size valid for all three granules at once -- silently gave NO cap for 13, 14, 15 and 17 .. 31 per CU:
    this test found it.)"""
    lib.agt_lk_lds_request.argtypes = [ctypes.c_int]
    lib.agt_lk_lds_request.restype = ctypes.c_int
    lds_cu = 160 * 1024
    up = lambda v, g: (v + g - 1) // g * g
    prev = None
    for n in range(2, 33):
        v = lib.agt_lk_lds_request(n)
        assert 0 < v <= 64 * 1024 and v % 256 == 0, (n, v)
        for g in (256, 512, 1024, 1280, 2048):
            assert up(v, g) * (n + 1) > lds_cu, (n, v, g)
        if n <= 16:                 # (registers allow 16 trackers per CU: the counts that can matter)
            for g in (256, 512):
                assert up(v, g) * n <= lds_cu, (n, v, g)
            for g in (1024, 1280):
                assert up(v, g) * (n - 1) <= lds_cu, (n, v, g)
        assert prev is None or v <= prev
        prev = v
    for n in (8, 10, 12):
        assert all(up(lib.agt_lk_lds_request(n), g) * n <= lds_cu for g in (1024, 1280))
    assert lib.agt_lk_lds_request(10) == 15104 and lib.agt_lk_lds_request(8) == 18432 and lib.agt_lk_lds_request(12) == 12800
    for n in (0, -1, 1):
        assert lib.agt_lk_lds_request(n) == 0
