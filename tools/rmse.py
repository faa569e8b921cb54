"""RMSE between two images of the same size (the gate "final images match within a stated RMSE at equal spp").
    python tools/rmse.py a.is b.is          image-state files (saveImageState): compared as mean radiance (sum / spp)
    python tools/rmse.py a.ppm b.ppm        8-bit P6 files: compared in [0, 1]
Prints one JSON line: {"rmse": ..., "max_abs": ..., "mean_a": ..., "mean_b": ..., "shape": [...]}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                      # noqa: E402


def read_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255", "not a binary 8-bit PPM"
    w, h = (int(x) for x in parts[1].split())
    return np.frombuffer(parts[3], np.uint8, w * h * 3).reshape(h, w, 3).astype(np.float64) / 255.0


def read_any(path):
    if path.endswith(".ppm"):
        return read_ppm(path)
    from craytracer_amd import api
    buf, ns = api.image_state_load(path)
    return buf.astype(np.float64) / max(ns, 1)


def rmse(a, b):
    assert a.shape == b.shape, "image sizes differ: %s vs %s" % (a.shape, b.shape)
    d = a - b
    return {"rmse": float(np.sqrt(np.mean(d * d))), "max_abs": float(np.abs(d).max()) if d.size else 0.0,
            "mean_a": float(a.mean()) if a.size else 0.0, "mean_b": float(b.mean()) if b.size else 0.0, "shape": list(a.shape)}


if __name__ == "__main__":
    print(json.dumps(rmse(read_any(sys.argv[1]), read_any(sys.argv[2]))))
