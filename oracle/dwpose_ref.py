"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy / plain Python) of the key-point rasterisation that turns SMGA's output into the
pose / mask frames of Stage 2 (second half).  Nothing under mmgt_amd/, scripts/ or bench.py's timed region may import it.

What it restates, with the reference lines:

  * `denormalize`, `mask_leg`, `process_keypoints`           data/extract_movment_mask_all.py:66-89,98-119,128-132
  * `DWposeDetector_movment_mask.__call__`                   src/dwpose/__init__.py:220-283 (one person per frame: max_ind = 0)
  * `draw_pose`, `draw_pose_mask_head / _lips / _hand`       src/dwpose/__init__.py:133-196
  * `draw_bodypose`                                          src/dwpose/util.py:79-157   (17 limbs as filled ellipse polygons, x 0.9, 18 joints)
  * `draw_handpose`                                          src/dwpose/util.py:160-206  (20 HSV-coloured edges of thickness 2, 21 joints per hand)
  * `draw_facepose`                                          src/dwpose/util.py:291-302  (68 white dots of radius 3, module eps = 0.01)
  * `draw_handpose_with_individual_bbox`, `draw_facepose_with_bbox`   src/dwpose/util.py:208-230,349-388

PARITY UNPINNED for the OpenCV part: the reference draws with cv2 (`ellipse2Poly`, `fillConvexPoly`, `line`, `circle`), which is not in this
image and not under /root/reference, and the reference holds no test or fixture of drawn frames.  The drawing primitives below restate
OpenCV 4.x's published algorithms (modules/imgproc/src/drawing.cpp: `ellipse2Poly` with its integer-degree sine table, `FillConvexPoly` with
XY_SHIFT = 16 edge stepping, `Line` = 8-connected `LineIterator` after `clipLine`, `Line2`, `ThickLine`, `Circle`); what IS checked here are the
properties those algorithms guarantee (tests/test_dwpose.py: symmetric discs of the known radius-3 / radius-4 row widths, Bresenham end points,
ellipse area and axes, clipping at the frame).  numpy >= 2 scalar semantics (float32 stays float32 under Python-float operands) are assumed for
the reference's arithmetic between the key points and the integer arguments of the cv2 calls; `matplotlib.colors.hsv_to_rgb` is restated too.

Frames are (H, W, 3) uint8 in the array's own channel order (the reference hands the arrays to PIL as RGB without a swap).
"""
import math

import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT

LIMB_SEQ = [[2, 3], [2, 6], [3, 4], [4, 5], [6, 7], [7, 8], [2, 9], [9, 10], [10, 11], [2, 12], [12, 13], [13, 14], [2, 1], [1, 15], [15, 17],
            [1, 16], [16, 18], [3, 17], [6, 18]]
BODY_COLORS = [[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0], [0, 255, 85], [0, 255, 170],
               [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255], [170, 0, 255], [255, 0, 255], [255, 0, 170], [255, 0, 85]]
HAND_EDGES = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 5), (5, 6), (6, 7), (7, 8), (0, 9), (9, 10), (10, 11), (11, 12), (0, 13), (13, 14), (14, 15),
              (15, 16), (0, 17), (17, 18), (18, 19), (19, 20)]

# OpenCV's SinTable: sin of the integer degrees 0 .. 450, seven decimals, stored as float
SIN_TABLE = np.array([round(math.sin(math.radians(i)), 7) for i in range(451)], dtype=np.float32)


def cv_round(x):
    """cvRound: round half to even (lrint)."""
    return int(np.rint(np.float64(x)))


def hsv_to_rgb(h, s, v):
    """matplotlib.colors.hsv_to_rgb for one colour."""
    i = int(h * 6.0)
    f = h * 6.0 - i
    p, q, t = v * (1.0 - s), v * (1.0 - s * f), v * (1.0 - s * (1.0 - f))
    return [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)][i % 6]


def hand_edge_colors():
    """util.py:178-183: HSV wheel over the 20 edges -> * 255 -> reversed -> astype(uint8) (truncation)."""
    out = []
    for i in range(len(HAND_EDGES)):
        rgb = np.array(hsv_to_rgb(i / len(HAND_EDGES), 1.0, 1.0), dtype=np.float64) * 255
        out.append(tuple(int(c) for c in rgb[::-1].astype(np.uint8)))
    return out


# ------------------------------------------------------------------------------------------------ OpenCV primitives
def clip_line(width, height, p1, p2):
    """cv::clipLine on integer (64-bit) points; returns (inside, p1, p2)."""
    x1, y1 = p1
    x2, y2 = p2
    right, bottom = width - 1, height - 1
    if width <= 0 or height <= 0:
        return False, p1, p2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def line8(img, p1, p2, color):
    """cv::Line with connectivity 8: clipLine + the 8-connected LineIterator (Bresenham), every visited pixel set."""
    h, w = img.shape[:2]
    ok, (x1, y1), (x2, y2) = clip_line(w, h, (int(p1[0]), int(p1[1])), (int(p2[0]), int(p2[1])))
    if not ok:
        return
    dx, dy = x2 - x1, y2 - y1
    sx = sy = 1
    if dx < 0:
        dx, sx = -dx, -1
    if dy < 0:
        dy, sy = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
    err = dx - (dy + dy)
    plus_delta, minus_delta = dx + dx, -(dy + dy)
    x, y = x1, y1
    for _ in range(dx + 1):
        img[y, x] = color
        mask = err < 0
        err += minus_delta + (plus_delta if mask else 0)
        if vert:                      # the major axis is y
            y += sy
            if mask:
                x += sx
        else:
            x += sx
            if mask:
                y += sy


def line2(img, p1, p2, color):
    """cv::Line2: a line between fixed-point (XY_SHIFT) end points, used for the outline of sub-pixel polygons."""
    h, w = img.shape[:2]
    ok, (x1, y1), (x2, y2) = clip_line(w << XY_SHIFT, h << XY_SHIFT, p1, p2)
    if not ok:
        return

    def put(x, y):
        if 0 <= x < w and 0 <= y < h:
            img[y, x] = color
    dx, dy = x2 - x1, y2 - y1
    ax, ay = abs(dx), abs(dy)
    if ax > ay:
        if dx < 0:
            dy = -dy
            x1, y1, x2, y2 = x2, y2, x1, y1
        x_step, y_step = XY_ONE, _cdiv(dy << XY_SHIFT, ax | 1)
        ecount = (x2 - x1) >> XY_SHIFT
    else:
        if dy < 0:
            dx = -dx
            x1, y1, x2, y2 = x2, y2, x1, y1
        x_step, y_step = _cdiv(dx << XY_SHIFT, ay | 1), XY_ONE
        ecount = (y2 - y1) >> XY_SHIFT
    x1 += XY_ONE >> 1
    y1 += XY_ONE >> 1
    put((x2 + (XY_ONE >> 1)) >> XY_SHIFT, (y2 + (XY_ONE >> 1)) >> XY_SHIFT)
    if ax > ay:
        x1 >>= XY_SHIFT
        while ecount >= 0:
            put(x1, y1 >> XY_SHIFT)
            x1 += 1
            y1 += y_step
            ecount -= 1
    else:
        y1 >>= XY_SHIFT
        while ecount >= 0:
            put(x1 >> XY_SHIFT, y1)
            x1 += x_step
            y1 += 1
            ecount -= 1


def _cdiv(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def ellipse2poly(center, axes, angle, arc_start=0, arc_end=360, delta=1):
    """cv::ellipse2Poly(Point, Size, int, int, int, int): the double-precision points rounded (cvRound), consecutive duplicates dropped."""
    while angle < 0:
        angle += 360
    while angle > 360:
        angle -= 360
    if arc_start > arc_end:
        arc_start, arc_end = arc_end, arc_start
    while arc_start < 0:
        arc_start += 360
        arc_end += 360
    while arc_end > 360:
        arc_end -= 360
        arc_start -= 360
    if arc_end - arc_start > 360:
        arc_start, arc_end = 0, 360
    beta, alpha = SIN_TABLE[angle], SIN_TABLE[450 - angle]          # sincos(angle, alpha = cos, beta = sin), floats
    pts, prev = [], None
    i = arc_start
    while i < arc_end + delta:
        a = min(i, arc_end)
        if a < 0:
            a += 360
        x = np.float64(axes[0]) * np.float64(SIN_TABLE[450 - a])
        y = np.float64(axes[1]) * np.float64(SIN_TABLE[a])
        px = np.float64(center[0]) + x * np.float64(alpha) - y * np.float64(beta)
        py = np.float64(center[1]) + x * np.float64(beta) + y * np.float64(alpha)
        pt = (cv_round(px), cv_round(py))
        if pt != prev:
            pts.append(pt)
            prev = pt
        i += delta
    if len(pts) == 1:
        pts = [tuple(center), tuple(center)]
    return pts


def fill_convex_poly(img, pts, color, shift=0):
    """cv::FillConvexPoly, line_type 8: the outline as lines, then the spans between the two edge chains (XY_SHIFT fixed point)."""
    h, w = img.shape[:2]
    n = len(pts)
    delta = (1 << shift) >> 1
    v = [(int(x), int(y)) for x, y in pts]
    xmin = xmax = v[0][0]
    ymin = ymax = v[0][1]
    imin = 0
    p0 = (v[-1][0] << (XY_SHIFT - shift), v[-1][1])
    for i, (px, py) in enumerate(v):
        if py < ymin:
            ymin, imin = py, i
        ymax, xmax, xmin = max(ymax, py), max(xmax, px), min(xmin, px)
        p = (px << (XY_SHIFT - shift), py)
        if shift == 0:
            line8(img, (p0[0] >> XY_SHIFT, p0[1]), (p[0] >> XY_SHIFT, p[1]), color)
        else:
            line2(img, p0, p, color)
        p0 = p
    xmin, xmax = (xmin + delta) >> shift, (xmax + delta) >> shift
    ymin, ymax = (ymin + delta) >> shift, (ymax + delta) >> shift
    if n < 3 or xmax < 0 or ymax < 0 or xmin >= w or ymin >= h:
        return
    ymax = min(ymax, h - 1)
    edge = [dict(idx=imin, di=1, x=-XY_ONE, dx=0, ye=ymin), dict(idx=imin, di=n - 1, x=-XY_ONE, dx=0, ye=ymin)]
    edges = n
    y = ymin
    while True:
        for e in edge:
            if y >= e["ye"]:
                idx0, di = e["idx"], e["di"]
                idx = idx0 + di
                if idx >= n:
                    idx -= n
                while True:
                    cont = edges > 0
                    edges -= 1
                    if not cont:
                        break
                    ty = (v[idx][1] + delta) >> shift
                    if ty > y:
                        xs, xe = v[idx0][0], v[idx][0]
                        if shift != XY_SHIFT:
                            xs <<= XY_SHIFT - shift
                            xe <<= XY_SHIFT - shift
                        e["ye"] = ty
                        e["dx"] = _cdiv((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e["x"] = xs
                        e["idx"] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= n:
                        idx -= n
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if edge[0]["x"] > edge[1]["x"] else (0, 1)
            xx1 = (edge[left]["x"] + (XY_ONE >> 1)) >> XY_SHIFT
            xx2 = (edge[right]["x"] + (XY_ONE >> 1)) >> XY_SHIFT
            if xx2 >= 0 and xx1 < w:
                img[y, max(xx1, 0):min(xx2, w - 1) + 1] = color
        edge[0]["x"] += edge[0]["dx"]
        edge[1]["x"] += edge[1]["dx"]
        y += 1
        if y > ymax:
            break


def circle_rows(radius):
    """The rows a filled cv::Circle covers: half width of the row at vertical distance |dy| (the midpoint loop of drawing.cpp)."""
    hw = [-1] * (radius + 1)
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        hw[dy] = max(hw[dy], dx)
        hw[dx] = max(hw[dx], dy)
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0          # (err <= 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return hw


def circle(img, center, radius, color):
    """cv::circle(..., thickness=-1) = Circle(img, center, radius, color, fill=1): clipped horizontal runs."""
    h, w = img.shape[:2]
    cx, cy = int(center[0]), int(center[1])
    for d, half in enumerate(circle_rows(radius)):
        for y in {cy - d, cy + d}:
            if 0 <= y < h:
                x0, x1 = max(cx - half, 0), min(cx + half, w - 1)
                if x0 <= x1:
                    img[y, x0:x1 + 1] = color


def thick_line(img, p1, p2, color, thickness=2):
    """cv::line(..., thickness > 1) = ThickLine: a quadrilateral in fixed point + a filled disc at both ends."""
    p0 = (int(p1[0]) << XY_SHIFT, int(p1[1]) << XY_SHIFT)
    pe = (int(p2[0]) << XY_SHIFT, int(p2[1]) << XY_SHIFT)
    inv = 1.0 / XY_ONE
    dx, dy = (p0[0] - pe[0]) * inv, (pe[1] - p0[1]) * inv
    r = dx * dx + dy * dy
    odd = thickness & 1
    th = thickness << (XY_SHIFT - 1)
    if abs(r) > np.finfo(np.float64).eps:
        r = (th + odd * XY_ONE * 0.5) / math.sqrt(r)
        dpx, dpy = cv_round(dy * r), cv_round(dx * r)
        quad = [(p0[0] + dpx, p0[1] + dpy), (p0[0] - dpx, p0[1] - dpy), (pe[0] - dpx, pe[1] - dpy), (pe[0] + dpx, pe[1] + dpy)]
        fill_convex_poly(img, quad, color, shift=XY_SHIFT)
    for p in (p0, pe):
        c = ((p[0] + (XY_ONE >> 1)) >> XY_SHIFT, (p[1] + (XY_ONE >> 1)) >> XY_SHIFT)
        circle(img, c, (th + (XY_ONE >> 1)) >> XY_SHIFT, color)


# ------------------------------------------------------------------------------------------------ the reference's drawing functions
def draw_bodypose(canvas, candidate, subset):
    """util.py:79-157.  candidate (18, 2) float32 in [0, 1] units, subset (1, 18): the joint's own index or -1."""
    H, W, _ = canvas.shape
    for i in range(17):
        for n in range(len(subset)):
            index = subset[n][np.array(LIMB_SEQ[i]) - 1]
            if -1 in index:
                continue
            Y = candidate[index.astype(int), 0] * float(W)
            X = candidate[index.astype(int), 1] * float(H)
            mX, mY = np.mean(X), np.mean(Y)
            length = ((X[0] - X[1]) ** 2 + (Y[0] - Y[1]) ** 2) ** 0.5
            angle = math.degrees(math.atan2(X[0] - X[1], Y[0] - Y[1]))
            polygon = ellipse2poly((int(mY), int(mX)), (int(length / 2), 4), int(angle), 0, 360, 1)
            fill_convex_poly(canvas, polygon, BODY_COLORS[i])
    canvas = (canvas * 0.9).astype(np.uint8)
    for i in range(18):
        for n in range(len(subset)):
            index = int(subset[n][i])
            if index == -1:
                continue
            x, y = candidate[index][0:2]
            circle(canvas, (int(x * W), int(y * H)), 4, BODY_COLORS[i])
    return canvas


def draw_handpose(canvas, all_hand_peaks, eps=1e-5):
    """util.py:160-206."""
    H, W, _ = canvas.shape
    colors = hand_edge_colors()
    for peaks in all_hand_peaks:
        peaks = np.asarray(peaks, dtype=np.float32)
        for idx, (a, b) in enumerate(HAND_EDGES):
            x1, y1 = int(peaks[a][0] * W), int(peaks[a][1] * H)
            x2, y2 = int(peaks[b][0] * W), int(peaks[b][1] * H)
            if x1 > eps and y1 > eps and x2 > eps and y2 > eps:
                thick_line(canvas, (x1, y1), (x2, y2), colors[idx], 2)
        for xn, yn in peaks:
            x, y = int(xn * W), int(yn * H)
            if x > eps and y > eps:
                circle(canvas, (x, y), 4, (0, 0, 255))
    return canvas


def draw_facepose(canvas, all_lmks, eps=0.01):
    """util.py:291-302 (module-level eps = 0.01)."""
    H, W, _ = canvas.shape
    for lmks in all_lmks:
        for x, y in np.array(lmks):
            x, y = int(x * W), int(y * H)
            if x > eps and y > eps:
                circle(canvas, (x, y), 3, (255, 255, 255))
    return canvas


def _bbox(points, W, H):
    min_x, min_y, max_x, max_y = W, H, 0, 0
    for x, y in np.array(points):
        x, y = int(x * W), int(y * H)
        if x > 0 and y > 0:
            min_x, min_y, max_x, max_y = min(min_x, x), min(min_y, y), max(max_x, x), max(max_y, y)
    return min_x, min_y, max_x, max_y


def draw_handpose_with_individual_bbox(canvas, all_hand_peaks):
    """util.py:208-230."""
    H, W, _ = canvas.shape
    for peaks in all_hand_peaks:
        x0, y0, x1, y1 = _bbox(peaks, W, H)
        if x0 < x1 and y0 < y1:
            canvas[y0:y1, x0:x1, :] = 255
    return canvas


def draw_facepose_with_bbox(canvas, all_lmks):
    """util.py:349-388: ONE box over all the landmark sets."""
    H, W, _ = canvas.shape
    box = None
    for lmks in all_lmks:
        x0, y0, x1, y1 = _bbox(lmks, W, H)
        if x0 < x1 and y0 < y1:
            box = [x0, y0, x1, y1] if box is None else [min(box[0], x0), min(box[1], y0), max(box[2], x1), max(box[3], y1)]
    if box:
        canvas[box[1]:box[3], box[0]:box[2], :] = 255
    return canvas


# ------------------------------------------------------------------------------------------------ key points -> the four frame streams
def denormalize(data):
    """extract_movment_mask_all.py:128-132 (float32 in, float32 arithmetic)."""
    data = (data + 1) / 2
    return data * (800 - (-200)) + (-200)


def mask_leg(kp402):
    """extract_movment_mask_all.py:66-89: key points 9, 10, 12, 13 (knees / ankles) zeroed, score included."""
    k = kp402.reshape(kp402.shape[0], 134, 3).copy()
    k[:, [9, 10, 12, 13], :] = 0
    return k.reshape(kp402.shape[0], -1)


def frame_streams(kp402_normalised, H=512, W=512):
    """`pose_vid_generator`'s frames without the mp4 files (extract_movment_mask_all.py:319-321 -> process_keypoints ->
    DWposeDetector_movment_mask.__call__): (pose, hands, lips, face) uint8 arrays of shape (L, H, W, 3); face = face box + hand boxes with
    uint8 wrap-around, as `detected_map_face + detected_map_hands` does (__init__.py:266)."""
    assert (H, W) == (512, 512), "the reference draws at 512 x 512 (its cv2.resize to 512 x 512 is then the identity)"
    rec = denormalize(np.asarray(kp402_normalised, dtype=np.float32))
    pose, hands_m, lips_m, face_m = [], [], [], []
    for row in rec:
        info = mask_leg(row[None]).reshape(1, 134, 3)
        candidate, subset = info[..., :2].copy(), info[..., 2].copy()
        candidate[..., 0] /= float(512)
        candidate[..., 1] /= float(512)
        score = subset[:, :18].copy()
        body = candidate[0, :18].copy()
        for j in range(18):
            score[0][j] = j if score[0][j] > 0.3 else -1
        candidate[subset < 0.3] = -1
        faces, lips = candidate[[0], 24:92], candidate[[0], 72:92]
        hands = np.vstack([candidate[[0], 92:113], candidate[[0], 113:]])
        z = lambda: np.zeros((H, W, 3), dtype=np.uint8)
        p = draw_facepose(draw_handpose(draw_bodypose(z(), body, score), hands), faces)
        hm = draw_handpose_with_individual_bbox(z(), hands)
        lm = draw_facepose_with_bbox(z(), lips)
        fm = draw_facepose_with_bbox(z(), faces) + hm          # uint8: 255 + 255 wraps to 254 where the boxes overlap
        pose.append(p)
        hands_m.append(hm)
        lips_m.append(lm)
        face_m.append(fm)
    return np.stack(pose), np.stack(hands_m), np.stack(lips_m), np.stack(face_m)
