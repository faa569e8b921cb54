"""Texel lookup and the environment map's lat-long mapping (textures + environment map), CPU side.

getTexColor (texture.cpp:27-86) cannot be compiled from the reference -- texture.cpp includes stb_image.h, which the
reference does not ship -- so the oracle's restatement is PARITY UNPINNED; it is checked here against an independent numpy
restatement of the same lines and on hand-computed texels.  cartesianToSpherical + sphericalToUV (util/math.h:95-107) are
header code and ARE pinned: the reference's own functions, compiled into oracle/_ref, against the oracle's (which takes the
double atan2 / acos rounded once where the reference calls atan2f / acosf -- the allowance is written in the test)."""
import ctypes as C

import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol


def _p(a):
    return a.ctypes.data


def tex_struct(img):
    t = api.CrtTexture()
    t.height, t.width, t.comp = img.shape
    t.is_float = 1 if img.dtype == np.float32 else 0
    t.data = img.ctypes.data
    return t


def numpy_tex_color(img, uv):
    """texture.cpp:27-86, line by line, in float32"""
    h, w, comp = img.shape
    out = []
    for k, size in ((0, w), (1, h)):
        x = np.float32(uv[k])
        if x < 0:
            x = np.float32(-x)
        if x > 1:
            x = np.float32(x - np.float32(int(x)))
            x = np.float32(np.float32(1.0) - x)
        if k == 1:
            x = np.float32(np.float32(1.0) - x)
        x = np.float32(x * np.float32(size))
        i = int(x) + 1 if np.float32(x - np.float32(int(x))) > np.float32(0.5) else int(x)
        out.append(i % size)
    u, v = out
    return (img[v, u, :3].astype(np.float32) / np.float32(255.0)).astype(np.float32)


@pytest.mark.parametrize("is_float", [False, True])
def test_tex_color_against_numpy_restatement(is_float):
    L = ol.oracle()
    img = scenes.procedural_texture(37, 23, seed=4, is_float=is_float)
    t = tex_struct(img)
    rng = np.random.default_rng(9)
    uvs = np.concatenate([rng.uniform(-2.5, 3.5, (4000, 2)), rng.uniform(0, 1, (4000, 2)),
                          [[0, 0], [1, 1], [0.5, 0.5], [1.0, 0.0], [-0.0, 1.0], [2.0, -3.0], [0.999999, 1e-7]]]).astype(np.float32)
    for uv in uvs:
        got = np.zeros(3, np.float32)
        L.oracle_trace_tex_color(C.addressof(t), _p(np.ascontiguousarray(uv)), _p(got))
        assert ol.same_bits(got, numpy_tex_color(img, uv)), uv


def test_tex_color_known_texels():
    """v is flipped (row 0 is the TOP of the image, v = 1), the fraction rounds up above .5, the index wraps modulo the size"""
    L = ol.oracle()
    img = np.zeros((4, 4, 3), np.uint8)
    img[..., 0] = np.arange(4)[None, :] * 10          # red = column
    img[..., 1] = np.arange(4)[:, None] * 10          # green = row
    t = tex_struct(img)

    def at(u, v):
        got = np.zeros(3, np.float32)
        L.oracle_trace_tex_color(C.addressof(t), _p(np.array([u, v], np.float32)), _p(got))
        return tuple(int(round(x * 255)) for x in got[:2])
    assert at(0.0, 1.0) == (0, 0)                      # top-left
    assert at(0.3, 1.0) == (10, 0)                     # 1.2 -> column 1
    assert at(0.4, 1.0) == (20, 0)                     # 1.6 -> rounds up to column 2
    assert at(1.0, 1.0) == (0, 0)                      # 4.0 -> 4 % 4
    assert at(0.0, 0.0) == (0, 0)                      # v = 0 -> flipped to 1.0 -> row 4 % 4 = 0
    assert at(0.0, 0.5) == (0, 20)                     # flipped 0.5 -> 2.0 -> row 2
    assert at(-0.3, 1.0) == at(0.3, 1.0)               # mirrored about 0
    assert at(1.25, 1.0) == at(0.75, 1.0)              # beyond 1: 1 - frac


@pytest.mark.ref
def test_env_uv_against_the_references_own_mapping():
    L, R = ol.oracle(), ol.ref()
    R.ref_env_uv.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(3)
    d = rng.standard_normal((20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.concatenate([d, [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, 1.5, 0]]]).astype(np.float32)
    same = 0
    for v in d:
        v = np.ascontiguousarray(v)
        a, b = np.zeros(2, np.float32), np.zeros(2, np.float32)
        L.oracle_trace_env_uv(_p(v), _p(a))
        R.ref_env_uv(_p(v), _p(b))
        # atan2f / acosf (glibc, float) against the double function rounded once: at most the last bit of phi / theta,
        # which the two float32 quotients keep within two ulps of a value in [0, 1]
        assert np.all(np.abs(a - b) <= 2.4e-7), (v, a, b)
        same += int(ol.same_bits(a, b))
    # glibc's atan2f / acosf differ from the correctly rounded value in 16 % / 8 % of these arguments (measured here), by one ulp
    assert same >= 0.8 * len(d), same / len(d)
