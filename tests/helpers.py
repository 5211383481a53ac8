"""Small array helpers shared by the tests."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits(a, b):
    """Bitwise equality of two f32 arrays, any NaN matching any NaN."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def full_frame_xy(width, height):
    ys, xs = np.mgrid[0:height, 0:width]
    return np.stack([xs.ravel(), ys.ravel()], 1).astype(np.uint32)
