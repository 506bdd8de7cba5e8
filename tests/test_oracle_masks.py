"""The oracle's mask restatement (binarize, 3x3-cross erosion, bit packing) against the NumPy
statements of the reference and hand-made known answers (cv2 is absent, so the
erosion is pinned by construction on small cases)."""
import warnings

import numpy as np
import pytest

from oracle import cpu_oracle as orc


VALUES = np.array([0.0, 0.25, 0.5, 0.50000006, 0.75, 0.999, 0.9999999, 1.0, 1.0000001, 1.5, 2.0, 200.0, 255.0,
                   255.5, 256.0, 257.0, 511.0, 512.0, 65536.0, -0.5, -0.999, -1.0, -2.0, -256.0, 1e9, -1e9],
                  np.float32)


def test_binarize_matches_numpy_casts():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)               # out-of-range float -> uint8 casts warn
        want_astype = (VALUES.astype(np.uint8) != 0)                   # V3:222 then V3:225 (> 0.5 on a uint8)
        want_v3 = ((VALUES * 255).astype(np.uint8) == 255)             # V3:87; survives /255.0 -> astype(uint8) as 1
        want_gt = VALUES > 0.5                                         # Same_color.py:125
    assert np.array_equal(orc.binarize_f32(VALUES, 0).astype(bool), want_astype)
    assert np.array_equal(orc.binarize_f32(VALUES, 1).astype(bool), want_v3)
    assert np.array_equal(orc.binarize_f32(VALUES, 2).astype(bool), want_gt)
    assert not orc.binarize_f32(np.array([np.nan], np.float32), 2)[0]


def test_v3_round_trip_of_binary_masks():
    """On YOLO's exact 0/1 masks all three rules agree (so V2/V3/Same_color see the same members)."""
    m = (np.random.default_rng(0).random((3, 16, 32)) < 0.5).astype(np.float32)
    a, b, c = (orc.binarize_f32(m, k) for k in (0, 1, 2))
    assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, m.astype(np.uint8))


def _erode_numpy(a):
    """3x3 MORPH_ELLIPSE = plus-shaped cross; outside the image does not constrain (cv2.erode's default border)."""
    p = np.pad(a, 1, constant_values=1)
    return p[1:-1, 1:-1] & p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]


def test_erode_known_answers():
    full = np.ones((5, 5), np.uint8)
    assert np.array_equal(orc.erode_cross3(full), full)                # border = +inf: a full image stays full
    hole = full.copy(); hole[2, 2] = 0
    want = full.copy(); want[2, 2] = want[1, 2] = want[3, 2] = want[2, 1] = want[2, 3] = 0
    assert np.array_equal(orc.erode_cross3(hole), want)                # a hole grows into a plus, not a 3x3 square
    corner = full.copy(); corner[0, 0] = 0
    want = full.copy(); want[0, 0] = want[0, 1] = want[1, 0] = 0
    assert np.array_equal(orc.erode_cross3(corner), want)
    plus = np.zeros((5, 5), np.uint8); plus[2, 1:4] = 1; plus[1:4, 2] = 1
    want = np.zeros((5, 5), np.uint8); want[2, 2] = 1
    assert np.array_equal(orc.erode_cross3(plus), want)                # the structuring element itself -> its centre
    sq = np.zeros((5, 5), np.uint8); sq[1:4, 1:4] = 1
    assert np.array_equal(orc.erode_cross3(sq), want)                  # 3x3 square -> centre
    line = np.zeros((3, 7), np.uint8); line[0] = 1
    assert np.array_equal(orc.erode_cross3(line), np.zeros_like(line)) # the row below constrains
    one_row = np.ones((1, 7), np.uint8); one_row[0, 3] = 0
    want = np.array([[1, 1, 0, 0, 0, 1, 1]], np.uint8)
    assert np.array_equal(orc.erode_cross3(one_row), want)             # H = 1: only left/right neighbours exist


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (9, 1), (2, 2), (17, 33), (64, 96)])
def test_erode_matches_numpy_shift_and(shape):
    a = (np.random.default_rng(shape[0] * 100 + shape[1]).random(shape) < 0.8).astype(np.uint8)
    assert np.array_equal(orc.erode_cross3(a), _erode_numpy(a))


@pytest.mark.parametrize("iters", [0, 1, 2, 4])
def test_pack_masks_bits_and_iterations(iters):
    rng = np.random.default_rng(7)
    M, H, W = 11, 23, 37
    masks = (rng.random((M, H, W)) < 0.85).astype(np.uint8) * rng.integers(1, 256, (M, H, W), dtype=np.uint8)
    lab = orc.pack_masks(masks, iters, H, W)
    for m in range(M):
        e = (masks[m] != 0).astype(np.uint8)
        for _ in range(iters):
            e = _erode_numpy(e)
        assert np.array_equal((lab >> m) & 1, e)
    assert not (lab >> M).any()


def test_pack_masks_limits():
    assert np.array_equal(orc.pack_masks(np.zeros((0, 4, 6), np.uint8), 0, 4, 6), np.zeros((4, 6), np.uint32))
    lab = orc.pack_masks(np.ones((32, 3, 3), np.uint8), 0)
    assert (lab == 0xFFFFFFFF).all()
    with pytest.raises(ValueError):
        orc.pack_masks(np.ones((33, 3, 3), np.uint8), 0)
