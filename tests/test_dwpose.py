"""Key points -> pose / mask frames (second half): the numpy oracle of the reference's DWPose drawing code (oracle/dwpose_ref.py,
cv2 restated: parity unpinned, see its header) against the properties the restated OpenCV algorithms guarantee, and the HIP rasteriser
(mmgt_dwpose_draw) against the oracle, bit for bit on uint8."""
import numpy as np
import pytest
import torch

from oracle import dwpose_ref as D


def _canvas(n=64):
    return np.zeros((n, n, 3), dtype=np.uint8)


def test_filled_circle_rows_and_symmetry():
    """cv::Circle's midpoint loop: the row half-widths of the radii the reference uses, a symmetric disc, clipping at the frame."""
    assert D.circle_rows(1) == [1, 0] and D.circle_rows(3) == [3, 2, 2, 0] and D.circle_rows(4) == [4, 3, 3, 2, 0]
    img = _canvas()
    D.circle(img, (20, 30), 4, (7, 8, 9))
    m = img[..., 0] == 7
    assert m.sum() == 9 + 2 * (7 + 7 + 5 + 1) and (img[m] == (7, 8, 9)).all()
    assert np.array_equal(m[:, 16:25], m[:, 16:25][:, ::-1]) and np.array_equal(m[26:35], m[26:35][::-1])
    img = _canvas()
    D.circle(img, (0, 63), 4, (1, 1, 1))                    # a corner: a quarter of the disc (+ the axes)
    assert (img[..., 0] == 1).sum() == 5 + 4 + 4 + 3 + 1
    D.circle(img, (-9, 5), 4, (1, 1, 1))                    # entirely outside: nothing, no error
    assert (img[..., 0] == 1).sum() == 17


def test_line_iterator_and_clipping():
    """8-connected LineIterator: max(|dx|, |dy|) + 1 pixels, both end points, one pixel per major step; clipLine keeps the visible part."""
    for p1, p2 in (((3, 4), (40, 11)), ((40, 11), (3, 4)), ((5, 50), (9, 2)), ((7, 7), (7, 7)), ((0, 0), (63, 63))):
        img = _canvas()
        D.line8(img, p1, p2, (1, 0, 0))
        m = img[..., 0] == 1
        assert m.sum() == max(abs(p1[0] - p2[0]), abs(p1[1] - p2[1])) + 1 and m[p1[1], p1[0]] and m[p2[1], p2[0]]
    ok, a, b = D.clip_line(64, 64, (-10, 20), (100, 20))
    assert ok and a == (0, 20) and b == (63, 20)
    ok, a, b = D.clip_line(64, 64, (-10, -10), (-5, 80))
    assert not ok
    ok, a, b = D.clip_line(64, 64, (10, -20), (30, 100))
    assert ok and a[1] == 0 and b[1] == 63 and 10 <= a[0] <= b[0] <= 30
    img = _canvas()
    D.line8(img, (-30, 10), (90, 40), (1, 0, 0))            # crosses the whole frame: one pixel per column
    assert (img[..., 0] == 1).sum(0).tolist() == [1] * 64


def test_ellipse_polygon_and_convex_fill():
    """ellipse2Poly: axis-aligned extents, point symmetry about the centre; fillConvexPoly: the filled limb is convex, contains the polygon's
    vertices and has the ellipse's area (+ its rim)."""
    pts = D.ellipse2poly((100, 90), (30, 4), 0)
    xs, ys = [p[0] for p in pts], [p[1] for p in pts]
    assert (min(xs), max(xs), min(ys), max(ys)) == (70, 130, 86, 94) and pts[0] == pts[-1] == (130, 90)
    for ang in (0, 30, 77, 90, 180, 271, 359, 360):
        pts = D.ellipse2poly((100, 90), (37, 4), ang)
        s = set(pts)
        assert all((200 - x, 180 - y) in s for x, y in pts), ang      # sin(a + 180) = -sin(a) holds exactly in the table
        img = np.zeros((200, 200, 3), dtype=np.uint8)
        D.fill_convex_poly(img, pts, (5, 5, 5))
        m = img[..., 0] == 5
        assert all(m[y, x] for x, y in pts)
        assert abs(m.sum() - np.pi * 37.5 * 4.5) < 40, (ang, m.sum())
        rows = [np.flatnonzero(r) for r in m if r.any()]
        assert all(len(r) == r[-1] - r[0] + 1 for r in rows)           # every row is one run
        cols = [np.flatnonzero(c) for c in m.T if c.any()]
        assert all(len(c) == c[-1] - c[0] + 1 for c in cols)
    assert D.ellipse2poly((5, 5), (0, 0), 0) == [(5, 5), (5, 5)]


def test_thick_line_covers_the_segment_with_width_two():
    img = _canvas()
    D.thick_line(img, (8, 8), (50, 8), (3, 3, 3), 2)
    m = img[..., 0] == 3
    assert m[8, 8:51].all() and m[7:10, 20].sum() >= 2 and not m[4, 20] and not m[12, 20]
    img = _canvas()
    D.thick_line(img, (10, 10), (40, 45), (3, 3, 3), 2)
    m = img[..., 0] == 3
    assert m[10, 10] and m[45, 40] and 2 * 46 <= m.sum() <= 4 * 47
    img = _canvas()
    D.thick_line(img, (20, 20), (20, 20), (3, 3, 3), 2)      # zero length: the two end discs of radius 1
    assert (img[..., 0] == 3).sum() == 5


def test_hand_edge_colours_are_the_hsv_wheel():
    want = [(0, 0, 255), (0, 76, 255), (0, 153, 255), (0, 229, 255), (0, 255, 203), (0, 255, 127), (0, 255, 51), (25, 255, 0), (102, 255, 0), (178, 255, 0), (255, 255, 0), (255, 178, 0), (255, 102, 0), (255, 25, 0), (255, 0, 50), (255, 0, 127), (255, 0, 204), (229, 0, 255), (152, 0, 255), (76, 0, 255)]
    assert D.hand_edge_colors() == want                      # (the table mmgt_amd/csrc/dwpose.hip carries)


def make_keypoints(seed, frames):
    """SMGA-like normalised features: joints around a skeleton with noise, some scores below the visibility threshold, some points off canvas."""
    rng = np.random.default_rng(seed)
    kp = np.zeros((frames, 134, 3), dtype=np.float32)
    centre = rng.uniform(-0.45, 0.05, (frames, 1, 2))
    kp[..., :2] = centre + rng.normal(0, 0.12, (frames, 134, 2))
    kp[:, 24:92, :2] = centre + np.array([0.0, -0.12]) + rng.normal(0, 0.03, (frames, 68, 2))          # a face cluster
    for h, off in ((92, (-0.15, 0.05)), (113, (0.15, 0.05))):
        kp[:, h:h + 21, :2] = centre + np.array(off) + rng.normal(0, 0.025, (frames, 21, 2))            # two hand clusters
    kp[..., 2] = rng.uniform(-0.5996, -0.5990, (frames, 134))                                          # scores: denormalised 0.2 .. 0.5 around the 0.3 threshold
    kp[:, :, 2] = np.where(rng.uniform(size=(frames, 134)) < 0.8, rng.uniform(-0.59, 0.5, (frames, 134)), kp[:, :, 2])
    if frames > 2:
        kp[1, 92:113, :2] = kp[1, 24:45, :2]                 # a hand over the face: the boxes overlap (255 + 255 wraps)
        kp[2, :18, :2] += 0.9                                # a body partly below / right of the canvas
        kp[2, 3, :2] = kp[2, 2, :2]                          # a zero-length limb
    return kp.astype(np.float32)


def test_frame_streams_follow_the_reference_rules():
    kp = make_keypoints(3, 3)
    pose, hands, lips, face = D.frame_streams(kp.reshape(3, -1))
    assert pose.shape == (3, 512, 512, 3) and hands.dtype == np.uint8
    assert set(np.unique(hands)) <= {0, 255} and set(np.unique(lips)) <= {0, 255} and set(np.unique(face)) <= {0, 254, 255}
    assert (face[1] == 254).any()                            # overlapping face and hand boxes: uint8 wrap-around (__init__.py:266)
    assert (face[0][hands[0][..., 0] == 255] != 0).all()     # face stream = face box + hand boxes
    # nothing is drawn for the masked legs: with ONLY the leg joints visible the body layer stays empty
    only_legs = kp[:1].copy()
    only_legs[..., 2] = -1.0
    only_legs[0, [9, 10, 12, 13], 2] = 0.5
    assert D.frame_streams(only_legs.reshape(1, -1))[0].sum() == 0
    # limb pixels carry colour * 0.9, joints the full colour
    vals = set(map(tuple, pose[0].reshape(-1, 3)[::7].tolist()))
    assert vals & {(229, 0, 0), (229, 76, 0), (0, 229, 0), (229, 153, 0)} and (255, 255, 255) in set(map(tuple, pose[0].reshape(-1, 3).tolist()))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,frames", [(0, 6), (1, 5), (2, 24)])
def test_hip_rasteriser_is_bit_exact_with_the_oracle(seed, frames):
    """mmgt_dwpose_draw through the C ABI against oracle/dwpose_ref.frame_streams: every byte of the four streams."""
    from mmgt_amd import hip
    kp = make_keypoints(seed, frames)
    want = D.frame_streams(kp.reshape(frames, -1))
    got = hip.dwpose_draw(torch.from_numpy(kp).cuda())
    torch.cuda.synchronize()
    for name, g, w in zip(("pose", "hands", "lips", "face"), got, want):
        g = g.cpu().numpy()
        w = w if name == "pose" else w[..., 0]
        if name != "pose":
            assert (want[("pose", "hands", "lips", "face").index(name)][..., 0] == want[("pose", "hands", "lips", "face").index(name)][..., 2]).all()
        bad = np.argwhere(g != w)
        assert bad.size == 0, f"{name}: {len(bad)} bytes differ, first at {bad[:5].tolist()}"
    a, b = hip.dwpose_draw(torch.from_numpy(kp).cuda()), got
    assert all(torch.equal(x, y) for x, y in zip(a, b))       # run-to-run identical (no ordering race between primitives)


@pytest.mark.gpu
def test_hip_rasteriser_rejects_other_canvas_sizes():
    from mmgt_amd import hip
    with pytest.raises(RuntimeError, match="512 x 512"):
        hip.dwpose_draw(torch.zeros((1, 134, 3), device="cuda"), H=256, W=256)
