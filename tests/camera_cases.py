"""Camera set-ups shared by the primary-ray tests and the golden generator."""
import numpy as np

CASES = {
    # name: (position, look, up, focal_pt_dist, thin_lens (focal_length, lens_radius) | None, fov, width, height)
    "cornell_pinhole": ((278.0, 273.0, 800.0), (278.0, 273.0, 0.0), (0.0, 1.0, 0.0), None, None, 0.6911, 512, 512),
    "oblique_pinhole": ((3.1, 2.2, 7.7), (-0.4, 0.9, 0.3), (0.05, 1.0, -0.02), 0.05, None, 1.1, 1920, 1080),
    "thin_lens": ((0.0, 2.5, 9.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 0.05, (8.0, 0.12), 0.9, 640, 360),
}


def samples(name, n):
    seed = sum(ord(c) for c in name)
    rng = np.random.default_rng(seed)
    return rng.random((n, 2), dtype=np.float32), rng.random((n, 2), dtype=np.float32)
