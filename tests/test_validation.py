"""crt_scene_create must refuse a scene description a kernel could not walk safely -- every index a
kernel follows is checked on the host first (include/crt_gpu.h, DESIGN.md "Boundary").  No GPU is
needed: validation comes before the first HIP call, so a description that passes validation ends
in CRT_E_NODEVICE / CRT_E_HIP here and a corrupted one in CRT_E_INVALID / CRT_E_TOOBIG."""
import copy
import ctypes as C

import numpy as np
import pytest

from craytracer_amd import api, scenes

INVALID, TOOBIG = -1, -5


def create(desc):
    h = C.c_void_p()
    rc = api.lib().crt_scene_create(0, C.byref(desc), C.byref(h))
    if rc == 0:                                    # a GPU is present: the scene was created
        api.lib().crt_scene_destroy(h)
    return rc


def clone(desc):
    d = api.CrtSceneDesc()
    C.memmove(C.byref(d), C.byref(desc), C.sizeof(d))
    return d


@pytest.fixture(scope="module")
def built():
    si = scenes.analytic_scene()
    tri = scenes.random_tris(300, seed=4)
    return {"bvh4": api.BuiltScene(tri, api.ACCEL_BVH4), "bvh": api.BuiltScene(tri, api.ACCEL_BVH),
            "grid": api.BuiltScene(tri, api.ACCEL_GRID), "mixed": api.BuiltScene(si, api.ACCEL_BVH4)}


def test_valid_descriptions_pass_validation(built):
    for b in built.values():
        assert create(b.desc) not in (INVALID, TOOBIG)


def test_header_fields(built):
    for field, value in (("accel", 9), ("num_obj", -1), ("num_non_grid_obj", -1), ("num_non_grid_obj", 10 ** 6)):
        d = clone(built["bvh4"].desc)
        setattr(d, field, value)
        assert create(d) == INVALID, field
    d = clone(built["bvh4"].desc)
    d.obj_id = None
    assert create(d) == INVALID
    d = clone(built["bvh4"].desc)
    d.nodes4 = None
    assert create(d) == INVALID


def test_object_table(built):
    b = built["mixed"]
    for arr_name, bad in (("obj_ref", 10 ** 6), ("obj_ref", -1), ("obj_type", 99), ("obj_type", -3)):
        arr = getattr(b, arr_name).copy()
        for pos in range(len(arr)):
            a = arr.copy()
            a[pos] = bad
            d = clone(b.desc)
            setattr(d, arr_name, a.ctypes.data)
            assert create(d) == INVALID, (arr_name, pos, bad)


def test_qbvh_nodes(built):
    b = built["bvh4"]
    nodes = b.nodes4.copy()
    n_nodes, n_obj = len(nodes), b.desc.num_obj
    rng = np.random.default_rng(1)
    inner = [(i, c) for i in range(n_nodes) for c in range(4) if nodes["type"][i][c] == -1]
    leaf = [(i, c) for i in range(n_nodes) for c in range(4) if nodes["type"][i][c] >= 0]
    cases = []
    for i, c in [inner[k] for k in rng.choice(len(inner), 12, replace=False)]:
        cases += [(i, c, "child", n_nodes), (i, c, "child", -1), (i, c, "child", i), (i, c, "child", 0)]   # out of range, self, back edge
    for i, c in [leaf[k] for k in rng.choice(len(leaf), 12, replace=False)]:
        cases += [(i, c, "child", n_obj), (i, c, "child", -1), (i, c, "type", 5), (i, c, "type", -2), (i, c, "type", 100)]
    for i, c, field, value in cases:
        a = nodes.copy()
        a[field][i][c] = value
        if field == "child" and a["type"][i][c] >= 0 and 0 <= value and value + int(a["type"][i][c]) <= n_obj:
            continue                                                   # still a valid leaf range
        d = clone(b.desc)
        d.nodes4 = a.ctypes.data
        assert create(d) == INVALID, (i, c, field, value)
    d = clone(b.desc)
    d.num_nodes4 = 0
    assert create(d) == INVALID


def test_qbvh_too_deep_for_the_stack():
    """a chain of inner nodes deeper than the traversal stack holds (3 * depth + 1 > 48)"""
    n = 40
    nodes = np.zeros(n, api.NODE4_DTYPE)
    nodes["bbox"][:, :12] = 0.0
    nodes["bbox"][:, 12:] = 1.0
    for i in range(n):
        nodes["type"][i] = (-1 if i + 1 < n else 1, 0, 0, 0)
        nodes["child"][i] = (i + 1 if i + 1 < n else 0, 0, 0, 0)
    tri = scenes.random_tris(1, seed=1)
    b = api.BuiltScene(tri, api.ACCEL_BVH4)
    d = clone(b.desc)
    d.nodes4 = nodes.ctypes.data
    d.num_nodes4 = n
    assert create(d) == TOOBIG
    d.num_nodes4 = 10                                                  # node 9 then points past the array
    assert create(d) == INVALID


def test_bvh2_nodes(built):
    b = built["bvh"]
    nodes = b.nodes2.copy()
    n_nodes, n_obj = len(nodes), b.desc.num_obj
    inner = np.flatnonzero(nodes["left"] >= 0)
    leaf = np.flatnonzero(nodes["left"] < 0)
    cases = [(i, f, v) for i in inner[:10] for f in ("left", "right") for v in (n_nodes, int(i), 0)] + \
            [(i, "right", v) for i in inner[:10] for v in (-1,)] + \
            [(i, "right", v) for i in leaf[:10] for v in (n_obj, -1)]
    for i, field, value in cases:
        a = nodes.copy()
        a[field][i] = value
        d = clone(b.desc)
        d.nodes2 = a.ctypes.data
        assert create(d) == INVALID, (i, field, value)


def test_grid_cells(built):
    b = built["grid"]
    cs, ci = b.cell_start.copy(), b.cell_items.copy()
    d = clone(b.desc)
    bad = cs.copy(); bad[0] = 1
    d.cell_start = bad.ctypes.data
    assert create(d) == INVALID
    d = clone(b.desc)
    bad = cs.copy(); k = int(np.flatnonzero(np.diff(cs) > 0)[0]); bad[k + 1] = bad[k] - 1
    d.cell_start = bad.ctypes.data
    assert create(d) == INVALID
    for v in (-1, b.desc.num_obj):
        d = clone(b.desc)
        bad = ci.copy(); bad[len(bad) // 2] = v
        d.cell_items = bad.ctypes.data
        assert create(d) == INVALID
    d = clone(b.desc)
    d.grid.nx += 1                                                     # nx * ny * nz != num_cells
    assert create(d) == INVALID
    d = clone(b.desc)
    d.cell_start = None
    assert create(d) == INVALID


def test_object_ids_must_be_a_permutation(built):
    """crt_hit.obj indexes the caller's per-object tables (materials, shading records): an id outside
    [0, num_obj) or one that appears twice is refused before anything reaches a kernel."""
    b = built["mixed"]
    n = b.desc.num_obj
    for pos, bad in ((0, n), (n // 2, -1), (n - 1, 10 ** 6), (1, int(b.obj_id[0]))):      # out of range x3, duplicate
        ids = b.obj_id.copy()
        ids[pos] = bad
        d = clone(b.desc)
        d.obj_id = ids.ctypes.data
        assert create(d) == INVALID, (pos, bad)
    ids = b.obj_id.copy()                                                               # any permutation is fine
    d = clone(b.desc)
    d.obj_id = ids[::-1].copy().ctypes.data
    assert create(d) not in (INVALID, TOOBIG)
