"""ctypes access to the CPU checker (oracle/libcrt_oracle.so) and, in the build container only,
to the reference's own code (oracle/_ref/libcrt_ref.so).  Test infrastructure: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os

import numpy as np

from craytracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "libcrt_oracle.so")
ORACLE_O0_PATH = os.path.join(ROOT, "oracle", "libcrt_oracle_O0.so")
_oracle_by_path = {}
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libcrt_ref.so")

_oracle = None
_ref = None
vp = C.c_void_p


def oracle(path=None):
    """The CPU checker; path=ORACLE_O0_PATH loads the -O0 build (baseline timing only)."""
    global _oracle
    if path is not None:
        if path not in _oracle_by_path:
            saved, _oracle = _oracle, None
            try:
                _oracle_by_path[path] = _load_oracle(path)
            finally:
                _oracle = saved
        return _oracle_by_path[path]
    if _oracle is None:
        _oracle = _load_oracle(ORACLE_PATH)
    return _oracle


def _load_oracle(path):
    global _oracle
    if True:
        L = C.CDLL(path)
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
        f3 = C.POINTER(C.c_float)
        L.oracle_camera_init.argtypes = [C.POINTER(api.CrtCamera), C.c_int, C.c_float, C.c_float]
        L.oracle_camera_init.restype = None
        L.oracle_camera_look_at.argtypes = [C.POINTER(api.CrtCamera), f3, f3, f3]
        L.oracle_camera_look_at.restype = None
        L.oracle_camera_set_focal_dist.argtypes = [C.POINTER(api.CrtCamera), C.c_float]
        L.oracle_camera_set_focal_dist.restype = None
        L.oracle_film_dimension.argtypes = [C.POINTER(api.CrtFilm), C.c_float, C.c_int, C.c_int, C.POINTER(api.CrtCamera)]
        L.oracle_film_dimension.restype = None
        L.oracle_primary_rays.argtypes = [C.POINTER(api.CrtCamera), C.POINTER(api.CrtFilm), C.c_uint, C.c_size_t, vp, vp, vp]
        L.oracle_primary_rays.restype = None
        L.oracle_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_rng.restype = C.c_float
        L.oracle_render.argtypes = [vp, C.POINTER(api.CrtRenderDesc), C.POINTER(api.CrtCamera), C.POINTER(api.CrtFilm),
                                    C.c_uint32, C.c_uint32, vp, C.c_int]
        L.oracle_trace_shade_rec.argtypes = [C.POINTER(api.CrtSceneInput), C.c_int, vp, C.c_float, vp]
        L.oracle_trace_shade_rec.restype = None
        L.oracle_trace_shade_rec_any.argtypes = [C.POINTER(api.CrtSceneInput), vp, vp, vp]
        L.oracle_trace_shade_rec_any.restype = None
        L.oracle_trace_shade_rec_mesh.argtypes = [C.POINTER(api.CrtSceneInput), C.POINTER(api.CrtMeshShading), C.c_int, vp, C.c_float,
                                                  C.c_float, C.c_float, vp]
        L.oracle_trace_shade_rec_mesh.restype = None
        for name in ("oracle_trace_disk", "oracle_trace_hemisphere"):
            getattr(L, name).argtypes = [vp, vp]
            getattr(L, name).restype = None
        L.oracle_trace_local_basis.argtypes = [vp, vp, vp]
        L.oracle_trace_local_basis.restype = None
        L.oracle_trace_lambertian.argtypes = [vp] * 7
        L.oracle_trace_lambertian.restype = C.c_float
        L.oracle_trace_mirror.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_int)]
        L.oracle_trace_mirror.restype = C.c_float
        L.oracle_trace_orennayar.argtypes = [vp, C.c_float, vp, vp, vp, vp, vp, vp, vp]
        L.oracle_trace_orennayar.restype = C.c_float
        L.oracle_trace_transparent.argtypes = [C.c_float, C.c_float, C.c_float, vp, vp, vp, vp, C.POINTER(C.c_int)]
        L.oracle_trace_transparent.restype = C.c_float
        L.oracle_trace_material_bsdf.argtypes = [vp, C.c_float, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int), vp]
        L.oracle_trace_material_bsdf.restype = C.c_float
        L.oracle_trace_beckmann.argtypes = [C.c_float, C.c_float, vp, vp, vp, vp, vp]
        L.oracle_trace_beckmann.restype = None
        L.oracle_trace_tex_color.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_trace_env_uv.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_trace_roughness_to_alpha.argtypes = [C.c_float]
        L.oracle_trace_roughness_to_alpha.restype = C.c_float
    return L


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
        f3 = C.POINTER(C.c_float)
        L.ref_camera_setup.argtypes = [C.POINTER(api.CrtCamera), C.c_int, C.c_float, C.c_float, C.c_float, f3, f3, f3]
        L.ref_camera_setup.restype = None
        L.ref_film_dimension.argtypes = [C.POINTER(api.CrtFilm), C.c_float, C.c_int, C.c_int, C.POINTER(api.CrtCamera)]
        L.ref_film_dimension.restype = None
        L.ref_primary_rays.argtypes = [C.POINTER(api.CrtCamera), C.POINTER(api.CrtFilm), C.c_uint, C.c_size_t, vp, vp, vp]
        L.ref_primary_rays.restype = None
        L.ref_shade_rec.argtypes = [vp, C.c_int, vp, vp, C.c_float]
        L.ref_shade_rec.restype = C.c_float
        L.ref_shade_rec_full.argtypes = [vp, C.c_int, vp, vp, C.c_float]
        L.ref_shade_rec_full.restype = C.c_float
        L.ref_mesh_triangle_shade_rec_full.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float]
        L.ref_mesh_triangle_shade_rec_full.restype = C.c_float
        for name in ("ref_map_sample_to_disk", "ref_map_sample_to_hemisphere"):
            getattr(L, name).argtypes = [vp, vp]
            getattr(L, name).restype = None
        L.ref_vec3_in_local_basis.argtypes = [vp, vp, vp]
        L.ref_vec3_in_local_basis.restype = None
        L.ref_lambertian_bsdf.argtypes = [vp] * 7
        L.ref_lambertian_bsdf.restype = C.c_float
        L.ref_mirror_bsdf.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_int)]
        L.ref_mirror_bsdf.restype = C.c_float
        L.ref_mesh_triangle_shade_rec.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float]
        L.ref_mesh_triangle_shade_rec.restype = C.c_float
        L.ref_bake_triangle.argtypes = [vp] * 7
        L.ref_bake_triangle.restype = None
        L.ref_orennayar_bsdf.argtypes = [vp, C.c_float, vp, vp, vp, vp, vp, vp, vp]
        L.ref_orennayar_bsdf.restype = C.c_float
        L.ref_transparent_bsdf.argtypes = [C.c_float, C.c_float, C.c_uint, vp, vp, vp, vp, C.POINTER(C.c_int)]
        L.ref_transparent_bsdf.restype = C.c_float
        L.ref_material_bsdf.argtypes = [vp, C.c_uint, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int), vp]
        L.ref_material_bsdf.restype = C.c_float
        L.ref_rand_float_after_srand.argtypes = [C.c_uint]
        L.ref_rand_float_after_srand.restype = C.c_float
        L.ref_beckmann.argtypes = [C.c_float, C.c_float, vp, vp, vp, vp, vp]
        L.ref_beckmann.restype = None
        L.ref_roughness_to_alpha.argtypes = [C.c_float]
        L.ref_roughness_to_alpha.restype = C.c_float
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
    def __init__(self, scene_input, accel=None, sort_mode=0, built=None, lib_path=None):
        """built: an api.BuiltScene to adopt (same tree as the GPU) instead of building from input."""
        L = oracle(lib_path)
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

    def render(self, desc, cam, flm, first_sample, num_samples, color_buffer=None, nthreads=8):
        """oracle_render: adds samples [first_sample, first_sample + num_samples) to color_buffer [h, w, 3]."""
        if color_buffer is None:
            rows = getattr(desc, "rows", None)
            h = rows[1] - rows[0] if rows else flm.frame_res_height
            color_buffer = np.zeros((h, flm.frame_res_width, 3), np.float32)
        assert color_buffer.dtype == np.float32 and color_buffer.flags.c_contiguous
        rc = self.L.oracle_render(self.h, C.byref(desc.c), C.byref(cam), C.byref(flm), int(first_sample), int(num_samples),
                                  color_buffer.ctypes.data, int(nthreads))
        assert rc == 0, "oracle_render: scene outside the restricted form"
        return color_buffer

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


# ---- primary rays ---------------------------------------------------------------------------
def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def struct_bytes(st):
    return bytes(memoryview(st))


def oracle_camera(position, look, up=(0.0, 1.0, 0.0), focal_pt_dist=None, thin_lens=None):
    """init -> optional setFocalDist -> lookAt, by the oracle's restatement."""
    L = oracle()
    cam = api.CrtCamera()
    L.oracle_camera_init(C.byref(cam), 1 if thin_lens else 0, *(thin_lens or (0.0, 0.0)))
    if focal_pt_dist is not None:
        L.oracle_camera_set_focal_dist(C.byref(cam), float(focal_pt_dist))
    L.oracle_camera_look_at(C.byref(cam), _f3(position), _f3(look), _f3(up))
    return cam


def ref_camera(position, look, up=(0.0, 1.0, 0.0), focal_pt_dist=None, thin_lens=None):
    """The same sequence by the reference's own camera.cpp."""
    cam = api.CrtCamera()
    fl, lr = thin_lens or (0.0, 0.0)
    ref().ref_camera_setup(C.byref(cam), 1 if thin_lens else 0, fl, lr, float(focal_pt_dist or -1.0), _f3(position), _f3(look), _f3(up))
    return cam


def film(which, cam, fov, w, h):
    f = api.CrtFilm()
    (oracle().oracle_film_dimension if which == "oracle" else ref().ref_film_dimension)(C.byref(f), float(fov), int(w), int(h), C.byref(cam))
    return f


def primary_rays(which, cam, flm, first_pixel, n, image_samples=None, lens_samples=None):
    rays = np.empty((n, 6), np.float32)
    fn = oracle().oracle_primary_rays if which == "oracle" else ref().ref_primary_rays
    a = None if image_samples is None else np.ascontiguousarray(image_samples, np.float32)
    b = None if lens_samples is None else np.ascontiguousarray(lens_samples, np.float32)
    fn(C.byref(cam), C.byref(flm), int(first_pixel), int(n), None if a is None else _p(a), None if b is None else _p(b), _p(rays))
    return rays
