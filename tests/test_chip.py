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
