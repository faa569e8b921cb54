"""The product's host builder (craytracer_amd/csrc/crt_build.cpp, through the C ABI) against the
oracle's restatement of the reference builders and against the reference structures in the golden
files.  CPU only."""
import os
import sys

import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402


def check(si):
    for accel in (api.ACCEL_NONE, api.ACCEL_GRID, api.ACCEL_BVH, api.ACCEL_BVH4):
        o = ol.OracleScene(si, accel, sort_mode=0)
        b = api.BuiltScene(si, accel)
        ids, nng = o.order()
        assert np.array_equal(ids, b.obj_id) and nng == b.desc.num_non_grid_obj
        if accel == api.ACCEL_BVH4:
            assert ol.same_bits(o.nodes4(), b.nodes4)
        elif accel == api.ACCEL_BVH:
            assert ol.same_bits(o.nodes2(), b.nodes2)
        elif accel == api.ACCEL_GRID and b.desc.num_obj - b.desc.num_non_grid_obj == 0:
            # the reference's resolution formula divides by zero here (undefined): the product
            # builds one empty cell inside the inverted box instead
            assert (b.desc.grid.nx, b.desc.grid.ny, b.desc.grid.nz, b.desc.grid.num_cells) == (1, 1, 1, 1)
            assert list(b.cell_start) == [0, 0] and len(b.cell_items) == 0
        elif accel == api.ACCEL_GRID:
            g, cs, ci = o.grid()
            assert bytes(g) == bytes(b.desc.grid)
            assert np.array_equal(cs, b.cell_start) and np.array_equal(ci, b.cell_items)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 6, 17, 30, 31, 32, 100, 1000])
def test_triangle_counts_around_the_leaf_and_sort_thresholds(n):
    check(scenes.random_tris(n, seed=n + 1, size=0.2))


def test_named_scenes():
    for make in (scenes.cornell29, scenes.analytic_scene, lambda: scenes.analytic_scene(False),
                 lambda: scenes.courtyard(3, 16, 32, seed=2)):
        check(make())


def test_equal_centroids_use_stable_order():
    """two triangles per quad share an AABB -> equal centroids: the documented builder-order rule"""
    t = scenes.cube_tris((0, 0, 0), 1.0)
    si = api.SceneInput(np.concatenate([t + np.float32(3 * k) for k in range(20)]))
    check(si)


def test_threaded_build_equals_serial_rule():
    si = scenes.courtyard(20, 64, 64, seed=3)           # 163 852 triangles: forks worker threads
    o = ol.OracleScene(si, api.ACCEL_BVH4, sort_mode=0)
    b = api.BuiltScene(si, api.ACCEL_BVH4)
    assert np.array_equal(o.order()[0], b.obj_id)
    assert ol.same_bits(o.nodes4(), b.nodes4)
    o2 = ol.OracleScene(si, api.ACCEL_BVH, sort_mode=0)
    b2 = api.BuiltScene(si, api.ACCEL_BVH)
    assert ol.same_bits(o2.nodes2(), b2.nodes2)


@pytest.mark.parametrize("name", sorted(mg.SCENES))
def test_against_reference_structures(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)
    si = mg.SCENES[name][0]()
    b4 = api.BuiltScene(si, api.ACCEL_BVH4)
    assert np.array_equal(b4.obj_id, g["order_bvh4"]) and ol.same_bits(b4.nodes4, g["nodes4"])
    b2 = api.BuiltScene(si, api.ACCEL_BVH)
    assert np.array_equal(b2.obj_id, g["order_bvh"]) and ol.same_bits(b2.nodes2, g["nodes2"])
    bg = api.BuiltScene(si, api.ACCEL_GRID)
    assert bytes(bg.desc.grid) == g["grid_hdr"].tobytes()
    assert np.array_equal(bg.cell_start, g["cell_start"]) and np.array_equal(bg.cell_items, g["cell_items"])


def test_invalid_inputs_are_rejected():
    import ctypes as C
    L = api.lib()
    si = scenes.analytic_scene()
    cin = si.c_struct()
    h = C.c_void_p()
    assert L.crt_build_scene(C.byref(cin), 7, C.byref(h)) == -1
    bad = si.obj_ref.copy()
    bad[0] = 10 ** 6
    cin.obj_ref = bad.ctypes.data
    assert L.crt_build_scene(C.byref(cin), api.ACCEL_BVH4, C.byref(h)) == -1
