"""ctypes access to the CPU checker (oracle/libcrt_oracle.so) and, in the build container only,
to the reference's own code (oracle/_ref/libcrt_ref.so).  Test infrastructure: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os

import numpy as np

from craytracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "libcrt_oracle.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libcrt_ref.so")

_oracle = None
_ref = None
vp = C.c_void_p


def oracle():
    global _oracle
    if _oracle is None:
        L = C.CDLL(ORACLE_PATH)
        L.oracle_scene_create.argtypes = [C.POINTER(api.CrtSceneInput), C.c_int, C.c_int]
        L.oracle_scene_create.restype = vp
        L.oracle_scene_from_desc.argtypes = [C.POINTER(api.CrtSceneDesc)]
        L.oracle_scene_from_desc.restype = vp
        L.oracle_scene_destroy.argtypes = [vp]
        L.oracle_scene_destroy.restype = None
        L.oracle_scene_order.argtypes = [vp, vp, C.POINTER(C.c_int32)]
        L.oracle_scene_nodes4.argtypes = [vp, vp]
        L.oracle_scene_nodes2.argtypes = [vp, vp]
        L.oracle_scene_grid.argtypes = [vp, C.POINTER(api.CrtGrid), vp, vp]
        L.oracle_intersect.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, C.POINTER(api.CrtCounters), C.c_int]
        L.oracle_intersect.restype = None
        L.oracle_shadow.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.POINTER(api.CrtCounters), C.c_int]
        L.oracle_shadow.restype = None
        L.oracle_intersect_object.argtypes = [vp, C.c_int, vp, vp]
        L.oracle_intersect_object.restype = C.c_float
        L.oracle_shadow_object.argtypes = [vp, C.c_int, vp]
        L.oracle_shadow_object.restype = C.c_float
        L.oracle_ray_aabb.argtypes = [vp, vp, vp]
        L.oracle_ray_aabb.restype = C.c_float
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_PATH)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_PATH)
        L.ref_scene_create.argtypes = [C.POINTER(api.CrtSceneInput), C.c_int]
        L.ref_scene_create.restype = vp
        L.ref_scene_order.argtypes = [vp, vp, C.POINTER(C.c_int32)]
        L.ref_scene_nodes4.argtypes = [vp, vp]
        L.ref_scene_nodes2.argtypes = [vp, vp]
        L.ref_scene_grid.argtypes = [vp, C.POINTER(api.CrtGrid), vp, vp]
        L.ref_intersect.argtypes = [vp, vp, C.c_size_t, vp]
        L.ref_shadow.argtypes = [vp, vp, vp, C.c_size_t, vp, vp]
        L.ref_intersect_object.argtypes = [vp, C.c_int, vp, vp]
        L.ref_intersect_object.restype = C.c_float
        L.ref_shadow_object.argtypes = [vp, C.c_int, vp]
        L.ref_shadow_object.restype = C.c_float
        L.ref_ray_aabb.argtypes = [vp, vp, vp]
        L.ref_ray_aabb.restype = C.c_float
        L.ref_ray_aabb4.argtypes = [vp, vp, vp, vp]
        L.ref_ray_aabb4.restype = None
        L.ref_sizes.argtypes = [C.POINTER(C.c_int)] * 4
        _ref = L
    return _ref


def _p(a):
    return a.ctypes.data if a is not None and a.size else None


class _SceneBase:
    prefix = ""

    def order(self):
        n = self.L_order(self.h, None, None)
        ids = np.empty(n, np.int32)
        nng = C.c_int32()
        self.L_order(self.h, _p(ids), C.byref(nng))
        return ids, nng.value

    def nodes4(self):
        n = self.L_nodes4(self.h, None)
        out = np.zeros(n, api.NODE4_DTYPE)
        self.L_nodes4(self.h, _p(out))
        return out

    def nodes2(self):
        n = self.L_nodes2(self.h, None)
        out = np.zeros(n, api.NODE2_DTYPE)
        self.L_nodes2(self.h, _p(out))
        return out

    def grid(self):
        g = api.CrtGrid()
        total = self.L_grid(self.h, C.byref(g), None, None)
        cs = np.zeros(g.num_cells + 1, np.int32)
        ci = np.zeros(max(total, 0), np.int32)
        self.L_grid(self.h, C.byref(g), _p(cs), _p(ci))
        return g, cs, ci


class OracleScene(_SceneBase):
    def __init__(self, scene_input, accel=None, sort_mode=0, built=None):
        """built: an api.BuiltScene to adopt (same tree as the GPU) instead of building from input."""
        L = oracle()
        self.L = L
        self.input = scene_input
        if built is not None:
            self.h = L.oracle_scene_from_desc(C.byref(built.desc))
        else:
            cin = scene_input.c_struct()
            self.h = L.oracle_scene_create(C.byref(cin), int(accel), int(sort_mode))
        assert self.h
        self.L_order, self.L_nodes4, self.L_nodes2, self.L_grid = \
            L.oracle_scene_order, L.oracle_scene_nodes4, L.oracle_scene_nodes2, L.oracle_scene_grid

    def intersect(self, rays, mode=api.TRAVERSE_EXACT, nthreads=1, counters=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        hits = np.empty(len(rays), api.HIT_DTYPE)
        c = api.CrtCounters()
        self.L.oracle_intersect(self.h, _p(rays), len(rays), _p(hits), int(mode), C.byref(c), int(nthreads))
        if counters:
            return hits, {"nodes_visited": c.nodes_visited, "prims_tested": c.prims_tested}
        return hits

    def shadow(self, rays, light_dist, nthreads=1, counters=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        ld = np.ascontiguousarray(light_dist, np.float32)
        t = np.empty(len(rays), np.float32)
        occ = np.empty(len(rays), np.uint8)
        c = api.CrtCounters()
        self.L.oracle_shadow(self.h, _p(rays), _p(ld), len(rays), _p(t), _p(occ), C.byref(c), int(nthreads))
        if counters:
            return t, occ.astype(bool), {"nodes_visited": c.nodes_visited, "prims_tested": c.prims_tested}
        return t, occ.astype(bool)

    def intersect_object(self, index, ray):
        ray = np.ascontiguousarray(ray, np.float32)
        hit = np.zeros(1, api.HIT_DTYPE)
        t = self.L.oracle_intersect_object(self.h, int(index), _p(ray), _p(hit))
        return np.float32(t), hit[0]

    def shadow_object(self, index, ray):
        ray = np.ascontiguousarray(ray, np.float32)
        return np.float32(self.L.oracle_shadow_object(self.h, int(index), _p(ray)))

    def close(self):
        if self.h:
            self.L.oracle_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RefScene(_SceneBase):
    """The reference's own structs/builders/traversal (build container only; never freed)."""

    def __init__(self, scene_input, accel):
        L = ref()
        self.L = L
        self.input = scene_input
        cin = scene_input.c_struct()
        self.h = L.ref_scene_create(C.byref(cin), int(accel))
        assert self.h
        self.L_order, self.L_nodes4, self.L_nodes2, self.L_grid = \
            L.ref_scene_order, L.ref_scene_nodes4, L.ref_scene_nodes2, L.ref_scene_grid

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        hits = np.empty(len(rays), api.HIT_DTYPE)
        rc = self.L.ref_intersect(self.h, _p(rays), len(rays), _p(hits))
        assert rc == 0, rc
        return hits

    def shadow(self, rays, light_dist):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        ld = np.ascontiguousarray(light_dist, np.float32)
        t = np.empty(len(rays), np.float32)
        occ = np.empty(len(rays), np.uint8)
        rc = self.L.ref_shadow(self.h, _p(rays), _p(ld), len(rays), _p(t), _p(occ))
        assert rc == 0, rc
        return t, occ.astype(bool)

    def intersect_object(self, index, ray):
        ray = np.ascontiguousarray(ray, np.float32)
        hit = np.zeros(1, api.HIT_DTYPE)
        t = self.L.ref_intersect_object(self.h, int(index), _p(ray), _p(hit))
        return np.float32(t), hit[0]

    def shadow_object(self, index, ray):
        ray = np.ascontiguousarray(ray, np.float32)
        return np.float32(self.L.ref_shadow_object(self.h, int(index), _p(ray)))


def libc_srand(seed):
    C.CDLL(None).srand(int(seed))


def same_bits(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()
