"""Synthetic stereo-rig + Kinect scene for the evaluator tests (no reference data exists for it: the reference reads its
Kinect captures from files that are not in its repository)."""
import numpy as np

KINECT_CAM = np.array([[594.21434211923247, 0, 339.30780975300314], [0, 591.04053696870778, 242.73913761751615], [0, 0, 1]], np.float64)


def raw_from_mm(mm):
    """Inverse of the raw-depth-to-millimetres formula, rounded to the nearest raw value."""
    return np.rint((3.3309495161 - 1000.0 / mm) / 0.0030711016).astype(np.int64)


def mm_from_raw(raw):
    return 1000.0 / (raw * -0.0030711016 + 3.3309495161)


def consistent_scene(rows=480, cols=640, seed=0):
    """The stereo camera IS the Kinect (R = I, T = 0, same intrinsics, baseline 120 mm): a smooth surface seen by both, so the
    errors are the raw depth's quantisation only."""
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float64)
    z = 1500.0 + 600.0 * np.sin(x / 90.0) * np.cos(y / 70.0) + 1.5 * x          # millimetres
    raw = raw_from_mm(z)
    fx, cx, cy = KINECT_CAM[0, 0], KINECT_CAM[0, 2], KINECT_CAM[1, 2]
    base = 120.0
    disp = (fx * base / mm_from_raw(raw)).astype(np.float32)                     # the disparity of the quantised depth
    # fy differs from fx in the Kinect's constants; Q has one focal length, so y is reprojected with fx and compared on fy:
    # keep the test's tolerance on x, z and the count, not on y
    q = np.array([[1, 0, 0, -cx], [0, 1, 0, -cy], [0, 0, 0, fx], [0, 0, 1.0 / base, 0]], np.float64)
    rng = np.random.default_rng(seed)
    n = 20000
    pts = np.stack([rng.integers(0, cols, n), rng.integers(0, rows, n)], 1).astype(np.int32)
    return raw.astype(np.int32), disp, pts, q, KINECT_CAM, np.eye(3), np.zeros(3)


def random_scene(rows, cols, n, seed, depth_shape=(480, 640)):
    """A rig beside the Kinect (small rotation, 25 mm offset), random disparities and raw depths with holes (2047) and a few
    very near values: exercises every branch, physically meaningless."""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.integers(0, cols, n), rng.integers(0, rows, n)], 1).astype(np.int32)
    disp = rng.integers(-2, 400, size=(rows, cols)).astype(np.float32) * 0.25
    raw = rng.integers(600, 1060, size=depth_shape).astype(np.int32)
    raw[rng.random(depth_shape) < 0.1] = 2047
    raw[rng.random(depth_shape) < 0.02] = rng.integers(0, 2048)
    raw[rng.random(depth_shape) < 0.01] = 1090                                     # beyond the formula's pole: negative millimetres
    a, b = 0.03, -0.02
    ra = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    rb = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    q = np.array([[1, 0, 0, -cols / 2.0], [0, 1, 0, -rows / 2.0], [0, 0, 0, 700.0], [0, 0, 1.0 / 120.0, 0]], np.float64)
    return raw, disp, pts, q, KINECT_CAM, ra @ rb, np.array([-25.0, 3.0, 8.0])
