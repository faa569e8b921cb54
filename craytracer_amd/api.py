"""ctypes binding of include/crt_gpu.h (the C ABI of libcrt_gpu.so).

This is the host-side mirror of the reference's scene-intersection interface
(`intersectTest` / `shadowIntersectTest`, /root/reference/intersect.h:365,:443 over a
`SceneObjects` handle, scene/scenedata.h:20-28), batched: a `Scene` is built once from a list
of objects in insertion order and answers closest-hit and any-hit queries for arrays of rays.

There is NO CPU fallback: if the HIP library is missing or no GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRT_GPU_LIB") or os.path.join(_HERE, "csrc", "libcrt_gpu.so")   # CRT_GPU_LIB: experiment builds

# enums (include/crt_gpu.h; same values as shapes/objecttype.h:4-18 and scene/scenedata.h:12-18)
SPHERE, PLANE, RECTANGLE, AABOX, TRIANGLE, GENERICOPENCYLINDER, DISK, GENERICTORUS, INSTANCED, COMPOUND, \
    FLAT_TRIANGLE, SMOOTH_TRIANGLE = range(12)
ACCEL_NONE, ACCEL_GRID, ACCEL_BVH, ACCEL_BVH4 = range(4)
ACCEL_NAMES = {"NONE": ACCEL_NONE, "GRID": ACCEL_GRID, "BVH": ACCEL_BVH, "BVH4": ACCEL_BVH4}
TRAVERSE_AUTO, TRAVERSE_PRUNED, TRAVERSE_EXACT = range(3)
BUILD_SAH = 1
NOALLOC_MAX_RAYS = 1 << 22
TRI_TYPES = (TRIANGLE, FLAT_TRIANGLE, SMOOTH_TRIANGLE)

K_EPSILON = np.float32(0.000007)
TMAX = np.float32(3.402823466e+38)

HIT_DTYPE = np.dtype([("t", "<f4"), ("obj", "<i4"), ("part", "<i4"), ("beta", "<f4"), ("gamma", "<f4")])
assert HIT_DTYPE.itemsize == 20


class CrtShape(C.Structure):
    _fields_ = [("type", C.c_int32), ("inner", C.c_int32), ("first_part", C.c_int32), ("num_parts", C.c_int32),
                ("p", C.c_float * 12), ("inv_xform", C.c_float * 16)]


SHAPE_DTYPE = np.dtype([("type", "<i4"), ("inner", "<i4"), ("first_part", "<i4"), ("num_parts", "<i4"),
                        ("p", "<f4", (12,)), ("inv_xform", "<f4", (16,))])
assert SHAPE_DTYPE.itemsize == C.sizeof(CrtShape) == 128


class CrtSceneInput(C.Structure):
    _fields_ = [("num_obj", C.c_int32), ("obj_type", C.c_void_p), ("obj_ref", C.c_void_p),
                ("tri_verts", C.c_void_p), ("num_tris", C.c_int32),
                ("shapes", C.c_void_p), ("num_shapes", C.c_int32)]


class CrtBvh4Node(C.Structure):
    _fields_ = [("bbox", C.c_float * 24), ("child", C.c_int32 * 4), ("type", C.c_int8 * 4),
                ("axis", C.c_uint8), ("pad", C.c_uint8 * 11)]


NODE4_DTYPE = np.dtype([("bbox", "<f4", (24,)), ("child", "<i4", (4,)), ("type", "i1", (4,)),
                        ("axis", "u1"), ("pad", "u1", (11,))])
NODE2_DTYPE = np.dtype([("bmin", "<f4", (3,)), ("bmax", "<f4", (3,)), ("left", "<i4"), ("right", "<i4")])
assert NODE4_DTYPE.itemsize == C.sizeof(CrtBvh4Node) == 128 and NODE2_DTYPE.itemsize == 32


class CrtGrid(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("num_cells", C.c_int32),
                ("wx", C.c_float), ("wy", C.c_float), ("wz", C.c_float),
                ("bmin", C.c_float * 3), ("bmax", C.c_float * 3)]


class CrtSceneDesc(C.Structure):
    _fields_ = [("accel", C.c_int32), ("num_obj", C.c_int32), ("num_non_grid_obj", C.c_int32),
                ("obj_type", C.c_void_p), ("obj_ref", C.c_void_p), ("obj_id", C.c_void_p),
                ("tri_verts", C.c_void_p), ("num_tris", C.c_int32),
                ("shapes", C.c_void_p), ("num_shapes", C.c_int32),
                ("nodes4", C.c_void_p), ("num_nodes4", C.c_int32),
                ("nodes2", C.c_void_p), ("num_nodes2", C.c_int32),
                ("grid", CrtGrid), ("cell_start", C.c_void_p), ("cell_items", C.c_void_p)]


class CrtCamera(C.Structure):
    _fields_ = [("type", C.c_int32), ("focal_pt_dist", C.c_float), ("focal_length", C.c_float), ("lens_radius", C.c_float),
                ("position", C.c_float * 3), ("x_axis", C.c_float * 3), ("y_axis", C.c_float * 3), ("z_axis", C.c_float * 3),
                ("focal_point", C.c_float * 3)]


class CrtFilm(C.Structure):
    _fields_ = [("fov", C.c_float), ("frame_length", C.c_float), ("frame_height", C.c_float), ("pixel_length", C.c_float),
                ("frame_res_width", C.c_int32), ("frame_res_height", C.c_int32)]


class CrtCounters(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64)]


MAT_MATTE, MAT_MIRROR, MAT_TRANSPARENT, MAT_EMISSIVE, MAT_PLASTIC, MAT_GLASS, MAT_METAL = 1, 2, 4, 8, 16, 32, 64   # MatType, materials.h:8-18
MATERIAL_DTYPE = np.dtype([("type", "<i4"), ("color", "<f4", (3,)), ("intensity", "<f4"), ("sigma", "<f4"),
                           ("ior_in", "<f4"), ("ior_out", "<f4"), ("ks", "<f4", (3,)), ("k", "<f4", (3,)),
                           ("uroughness", "<f4"), ("vroughness", "<f4")])
assert MATERIAL_DTYPE.itemsize == 64
# procMetalType's table (materials.cpp:5-20): name -> (eta, k)
METALS = {
    "GOLD": ((0.14282006, 0.37414363, 1.43944442), (3.90463543, 2.44763327, 2.13765264)),
    "SILVER": ((0.154935181, 0.116475478, 0.138087392), (4.81810093, 3.11561656, 2.1424017)),
    "BERYLLIUM": ((4.17617416, 3.1783011, 2.77819276), (3.82729554, 3.00373626, 2.86292768)),
    "CHROMIUM": ((4.36040831, 2.9105196, 1.65118635), (5.19538164, 4.22238398, 3.74699736)),
    "CESIUM": ((2.14034843, 1.69870293, 1.65889668), (0.0, 0.0, 0.0)),
    "COPPER": ((0.19999069, 0.92208463, 1.09987593), (3.90463543, 2.44763327, 2.13765264)),
    "MERCURY": ((2.39383841, 1.43696785, 0.907622635), (6.31419611, 4.36266136, 3.41453838)),
}


class CrtMeshShading(C.Structure):
    _fields_ = [("flat_normal", C.c_void_p), ("vertex_normals", C.c_void_p), ("vertex_tangents", C.c_void_p),
                ("normal_mat_index", C.c_void_p), ("normal_mats", C.c_void_p), ("num_normal_mats", C.c_int32),
                ("vertex_uv", C.c_void_p)]


class CrtMesh(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("tangents", C.c_void_p), ("indices", C.c_void_p),
                ("num_vertices", C.c_int32), ("num_indices", C.c_int32)]


class CrtObjShape(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("texcoords", C.POINTER(C.c_float)),
                ("indices", C.POINTER(C.c_int32)), ("num_positions", C.c_int32), ("num_normals", C.c_int32),
                ("num_texcoords", C.c_int32), ("num_indices", C.c_int32), ("mat_name", C.c_char * 64)]


class CrtMeshEntry(C.Structure):
    _fields_ = [("scaling", C.c_float * 3), ("orientation", C.c_float * 3), ("location", C.c_float * 3), ("smooth", C.c_int32)]


class CrtRenderDesc(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("obj_material", C.c_void_p), ("materials", C.c_void_p), ("num_materials", C.c_int32),
                ("has_env_light", C.c_int32), ("env_color", C.c_float * 3), ("env_intensity", C.c_float),
                ("env_world_radius", C.c_float), ("max_depth", C.c_int32), ("samples_in_flight", C.c_uint32),
                ("seed", C.c_uint64), ("mesh_shading", C.c_void_p),
                ("textures", C.c_void_p), ("num_textures", C.c_int32), ("material_diffuse_tex", C.c_void_p),
                ("env_map", C.c_int32), ("env_transform", C.c_float * 9), ("row_begin", C.c_int32), ("row_end", C.c_int32)]


class CrtTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("comp", C.c_int32), ("is_float", C.c_int32), ("data", C.c_void_p)]


class CrtRenderStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("ms_total", C.c_double), ("ms_extend", C.c_double), ("ms_shadow", C.c_double)]


SR_HIT_POINT, SR_NORMAL, SR_DPDU, SR_DPDV, SR_WO, SR_UV, SR_BY_CONVENTION = 1, 2, 4, 8, 16, 32, 256
SHADE_REC_DTYPE = np.dtype([("hit_point", "<f4", (3,)), ("normal", "<f4", (3,)), ("dpdu", "<f4", (3,)), ("dpdv", "<f4", (3,)),
                            ("wo", "<f4", (3,)), ("uv", "<f4", (2,)), ("material", "<i4"), ("written", "<u4")])
assert SHADE_REC_DTYPE.itemsize == 76

# every symbol include/crt_gpu.h declares
EXPORTS = [
    "crt_scene_file_load", "crt_scene_file_get", "crt_scene_file_free", "crt_env_world_radius",
    "crt_build_scene", "crt_build_scene_ex", "crt_built_scene_desc", "crt_built_scene_destroy",
    "crt_device_count", "crt_scene_create", "crt_scene_destroy",
    "crt_scene_set_traverse_mode", "crt_scene_get_traverse_mode", "crt_scene_set_ray_binning", "crt_scene_get_ray_binning", "crt_scene_set_node_compression",
    "crt_host_grid_distance_field", "crt_host_qbvh_node_order", "crt_host_bvh2_wide_records",
    "crt_scene_set_wide_nodes", "crt_host_qbvh_wide8_records",
    "crt_host_alloc", "crt_host_free",
    "crt_camera_init", "crt_camera_look_at", "crt_camera_set_focal_dist", "crt_film_dimension", "crt_primary_rays_dev",
    "crt_intersect_batch", "crt_shadow_batch", "crt_intersect_batch_dev", "crt_shadow_batch_dev", "crt_intersect_batch_dev_noalloc", "crt_intersect_batch_dev_slice", "crt_shadow_batch_dev_slice", "crt_scene_reserve_rays", "crt_intersect_batch_dev_counted", "crt_shadow_batch_dev_counted", "crt_fill_shade_rec",
    "crt_scene_enable_counters", "crt_scene_get_counters", "crt_scene_reset_counters",
    "crt_scene_last_kernel_ms", "crt_scene_last_binning_ms", "crt_strerror", "crt_last_hip_error",
    "crt_scene_device", "crt_scene_num_objects", "crt_renderer_create", "crt_renderer_destroy", "crt_renderer_render", "crt_renderer_read",
    "crt_renderer_get_stats", "crt_renderer_read_dev", "crt_renderer_reset",
    "crt_bake_mesh", "crt_obj_load", "crt_obj_free", "crt_mesh_normals_tangents", "crt_image_state_save", "crt_image_state_load", "crt_image_state_free", "crt_image_from_color_buffer", "crt_ppm_write",
    "crt_texture_load_ppm", "crt_texture_free",
]

_lib = None


def lib():
    """Load libcrt_gpu.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "craytracer_amd: %s is missing -- build it with `make -C craytracer_amd/csrc` "
            "(there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  If the process
    # is going to use torch as well, torch's copy must be the one that gets loaded (same soname,
    # so libcrt_gpu.so then binds to it); loading /opt/rocm's first leaves torch without devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.crt_build_scene.argtypes = [C.POINTER(CrtSceneInput), C.c_int, C.POINTER(vp)]
    L.crt_build_scene_ex.argtypes = [C.POINTER(CrtSceneInput), C.c_int, C.c_int, C.POINTER(vp)]
    L.crt_built_scene_desc.argtypes = [vp]
    L.crt_built_scene_desc.restype = C.POINTER(CrtSceneDesc)
    L.crt_built_scene_destroy.argtypes = [vp]
    L.crt_built_scene_destroy.restype = None
    L.crt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.crt_scene_create.argtypes = [C.c_int, C.POINTER(CrtSceneDesc), C.POINTER(vp)]
    L.crt_scene_destroy.argtypes = [vp]
    L.crt_scene_destroy.restype = None
    L.crt_scene_set_traverse_mode.argtypes = [vp, C.c_int]
    L.crt_scene_get_traverse_mode.argtypes = [vp, C.POINTER(C.c_int)]
    L.crt_scene_set_ray_binning.argtypes = [vp, C.c_int]
    L.crt_scene_get_ray_binning.argtypes = [vp, C.POINTER(C.c_int)]
    L.crt_scene_set_node_compression.argtypes = [vp, C.c_int]
    L.crt_host_grid_distance_field.argtypes = [C.POINTER(CrtSceneDesc), vp]
    L.crt_host_qbvh_node_order.argtypes = [C.POINTER(CrtSceneDesc), C.c_int, vp, C.POINTER(C.c_int32)]
    L.crt_host_bvh2_wide_records.argtypes = [C.POINTER(CrtSceneDesc), vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.crt_scene_set_wide_nodes.argtypes = [vp, C.c_int]
    L.crt_host_qbvh_wide8_records.argtypes = [C.POINTER(CrtSceneDesc), vp, C.POINTER(C.c_int32)]
    f3 = C.POINTER(C.c_float)
    L.crt_camera_init.argtypes = [C.POINTER(CrtCamera), C.c_int, C.c_float, C.c_float]
    L.crt_camera_look_at.argtypes = [C.POINTER(CrtCamera), f3, f3, f3]
    L.crt_camera_set_focal_dist.argtypes = [C.POINTER(CrtCamera), C.c_float]
    L.crt_film_dimension.argtypes = [C.POINTER(CrtFilm), C.c_float, C.c_int, C.c_int, C.POINTER(CrtCamera)]
    L.crt_primary_rays_dev.argtypes = [C.c_int, C.POINTER(CrtCamera), C.POINTER(CrtFilm), C.c_uint32, C.c_size_t, vp, vp, vp, vp]
    L.crt_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.crt_host_free.argtypes = [vp]
    L.crt_host_free.restype = None
    L.crt_intersect_batch.argtypes = [vp, vp, C.c_size_t, vp]
    L.crt_shadow_batch.argtypes = [vp, vp, vp, C.c_size_t, vp, vp]
    L.crt_intersect_batch_dev.argtypes = [vp, vp, C.c_size_t, vp, vp]
    L.crt_shadow_batch_dev.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, vp]
    L.crt_intersect_batch_dev_noalloc.argtypes = [vp, vp, C.c_size_t, vp, vp]
    L.crt_intersect_batch_dev_counted.argtypes = [vp, vp, vp, C.c_size_t, vp, vp]
    L.crt_intersect_batch_dev_slice.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, vp]
    L.crt_shadow_batch_dev_slice.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.c_uint32, C.c_uint32, vp]
    L.crt_scene_reserve_rays.argtypes = [vp, C.c_size_t]
    L.crt_shadow_batch_dev_counted.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, vp, vp]
    L.crt_fill_shade_rec.argtypes = [C.POINTER(CrtSceneInput), vp, vp, vp, vp, C.c_size_t, C.c_int, vp]
    L.crt_scene_enable_counters.argtypes = [vp, C.c_int]
    L.crt_scene_get_counters.argtypes = [vp, C.POINTER(CrtCounters)]
    L.crt_scene_reset_counters.argtypes = [vp]
    L.crt_scene_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.crt_scene_last_binning_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.crt_scene_device.argtypes = [vp, C.POINTER(C.c_int)]
    L.crt_scene_num_objects.argtypes = [vp, C.POINTER(C.c_int)]
    L.crt_renderer_create.argtypes = [vp, C.POINTER(CrtRenderDesc), C.POINTER(CrtCamera), C.POINTER(CrtFilm), C.POINTER(vp)]
    L.crt_renderer_destroy.argtypes = [vp]
    L.crt_renderer_destroy.restype = None
    L.crt_renderer_render.argtypes = [vp, C.c_uint32]
    L.crt_renderer_read.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    L.crt_renderer_get_stats.argtypes = [vp, C.POINTER(CrtRenderStats)]
    L.crt_renderer_read_dev.argtypes = [vp, vp, vp]
    L.crt_renderer_reset.argtypes = [vp, C.c_uint32]
    L.crt_bake_mesh.argtypes = [C.POINTER(CrtMesh), C.POINTER(CrtMeshEntry), vp, vp, vp, vp, vp]
    L.crt_obj_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(CrtObjShape)), C.POINTER(C.c_int32)]
    L.crt_obj_free.argtypes = [C.POINTER(CrtObjShape), C.c_int32]
    L.crt_obj_free.restype = None
    L.crt_mesh_normals_tangents.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp, vp]
    L.crt_image_state_save.argtypes = [C.c_char_p, vp, C.c_int, C.c_int, C.c_int]
    L.crt_image_state_load.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.crt_image_state_free.argtypes = [vp]
    L.crt_image_state_free.restype = None
    L.crt_image_from_color_buffer.argtypes = [vp, vp, C.c_int, C.c_int]
    L.crt_ppm_write.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]
    L.crt_texture_load_ppm.argtypes = [C.c_char_p, C.POINTER(CrtTexture)]
    L.crt_texture_free.argtypes = [C.POINTER(CrtTexture)]
    L.crt_texture_free.restype = None
    L.crt_strerror.argtypes = [C.c_int]
    L.crt_strerror.restype = C.c_char_p
    L.crt_last_hip_error.argtypes = []
    L.crt_last_hip_error.restype = C.c_char_p
    _lib = L
    return L


class CrtError(RuntimeError):
    def __init__(self, code, where):
        L = lib()
        msg = "%s: %s (%d)" % (where, L.crt_strerror(code).decode(), code)
        hip = L.crt_last_hip_error().decode()
        if hip:
            msg += " [" + hip + "]"
        super().__init__(msg)
        self.code = code


def _check(code, where):
    if code != 0:
        raise CrtError(code, where)


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def make_camera(position, look_point, up=(0.0, 1.0, 0.0), focal_pt_dist=None, thin_lens=None, L=None):
    """Camera set up as the scene loader does (scenefile.h camera block): init, optional focal
    distance, look-at.  thin_lens = (focal_length, lens_radius) or None for a pinhole.
    L: library to call (default: the GPU library; tests pass the oracle / reference build)."""
    L = L or lib()
    cam = CrtCamera()
    if thin_lens:
        _check(L.crt_camera_init(C.byref(cam), 1, float(thin_lens[0]), float(thin_lens[1])), "crt_camera_init")
    else:
        _check(L.crt_camera_init(C.byref(cam), 0, 0.0, 0.0), "crt_camera_init")
    if focal_pt_dist is not None:
        _check(L.crt_camera_set_focal_dist(C.byref(cam), float(focal_pt_dist)), "crt_camera_set_focal_dist")
    _check(L.crt_camera_look_at(C.byref(cam), _f3(position), _f3(look_point), _f3(up)), "crt_camera_look_at")
    return cam


def make_film(cam, fov, width, height, L=None):
    L = L or lib()
    film = CrtFilm()
    _check(L.crt_film_dimension(C.byref(film), float(fov), int(width), int(height), C.byref(cam)), "crt_film_dimension")
    return film


def primary_rays_dev(cam, film, first_pixel, n, d_rays, d_image_samples=None, d_lens_samples=None, device=0, stream=None):
    _check(lib().crt_primary_rays_dev(device, C.byref(cam), C.byref(film), first_pixel, n, d_image_samples, d_lens_samples,
                                      d_rays, stream), "crt_primary_rays_dev")


class HostBuffer:
    """Page-locked host array from crt_host_alloc, viewed as numpy (`.array`); free with close()."""

    def __init__(self, shape, dtype):
        self._dtype = np.dtype(dtype)
        self._shape = tuple(np.atleast_1d(shape))
        nbytes = int(np.prod(self._shape)) * self._dtype.itemsize
        self._p = C.c_void_p()
        _check(lib().crt_host_alloc(nbytes, C.byref(self._p)), "crt_host_alloc")
        buf = (C.c_char * max(nbytes, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self._dtype, count=int(np.prod(self._shape))).reshape(self._shape)

    def close(self):
        if self._p is not None and self._p.value:
            self.array = None
            lib().crt_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SceneInput:
    """Objects in insertion order (what SceneObjects_push_obj accumulates, scenedata.h:40-48)."""

    def __init__(self, tri_verts=None, shapes=None, obj_type=None, obj_ref=None):
        self.tri_verts = np.ascontiguousarray(
            np.zeros((0, 9), np.float32) if tri_verts is None else tri_verts, dtype=np.float32).reshape(-1, 9)
        self.shapes = np.ascontiguousarray(
            np.zeros(0, SHAPE_DTYPE) if shapes is None else shapes, dtype=SHAPE_DTYPE)
        if obj_type is None:
            self.obj_type = None
            self.obj_ref = None
            self.num_obj = len(self.tri_verts)
        else:
            self.obj_type = np.ascontiguousarray(obj_type, dtype=np.int32)
            self.obj_ref = np.ascontiguousarray(obj_ref, dtype=np.int32)
            self.num_obj = len(self.obj_type)

    def c_struct(self):
        s = CrtSceneInput()
        s.num_obj = self.num_obj
        s.obj_type = _ptr(self.obj_type) if self.obj_type is not None else None
        s.obj_ref = _ptr(self.obj_ref) if self.obj_ref is not None else None
        s.tri_verts = _ptr(self.tri_verts)
        s.num_tris = len(self.tri_verts)
        s.shapes = _ptr(self.shapes)
        s.num_shapes = len(self.shapes)
        return s


class RenderDesc:
    """Materials, lights and limits of the restricted path tracer (crt_render_desc).  Keeps the numpy
    arrays alive; `.c` is the ctypes struct."""

    def __init__(self, scene_input, obj_material, materials, max_depth, seed=0, env=None, samples_in_flight=0, mesh_shading=None,
                 textures=None, material_diffuse_tex=None, env_map=None, env_transform=None, rows=None):
        """textures: list of [h, w, 3] arrays (uint8, or float32 for an EXR-style map); material_diffuse_tex: per material an
        index into textures or -1; env_map: index of the environment map (needs env); env_transform: 3 x 3; rows: (begin, end)
        image rows this renderer renders (tile sharding)."""
        self.scene_input = scene_input
        self._in = scene_input.c_struct()
        self.obj_material = np.ascontiguousarray(obj_material, dtype=np.int32)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        assert len(self.obj_material) == scene_input.num_obj
        c = CrtRenderDesc()
        c.objects = C.addressof(self._in)
        c.obj_material = _ptr(self.obj_material)
        c.materials = _ptr(self.materials)
        c.num_materials = len(self.materials)
        if env is not None:                       # (color[3], intensity, world_radius)
            c.has_env_light = 1
            c.env_color = (C.c_float * 3)(*[float(x) for x in env[0]])
            c.env_intensity = float(env[1])
            c.env_world_radius = float(env[2])
        c.max_depth = int(max_depth)
        c.samples_in_flight = int(samples_in_flight)
        c.seed = int(seed)
        self.mesh_shading = mesh_shading                  # a MeshShading (keeps its arrays alive) or None
        if mesh_shading is not None:
            c.mesh_shading = C.addressof(mesh_shading.c)
        self.textures = []
        if textures:
            self._tex = (CrtTexture * len(textures))()
            for k, t in enumerate(textures):
                t = np.ascontiguousarray(t)
                assert t.ndim == 3 and t.shape[2] in (3, 4) and t.dtype in (np.uint8, np.float32)
                self.textures.append(t)
                self._tex[k].height, self._tex[k].width, self._tex[k].comp = t.shape
                self._tex[k].is_float = 1 if t.dtype == np.float32 else 0
                self._tex[k].data = t.ctypes.data
            c.textures = C.addressof(self._tex)
            c.num_textures = len(textures)
        self.material_diffuse_tex = None
        if material_diffuse_tex is not None:
            self.material_diffuse_tex = np.ascontiguousarray(material_diffuse_tex, dtype=np.int32)
            assert len(self.material_diffuse_tex) == len(self.materials)
            c.material_diffuse_tex = _ptr(self.material_diffuse_tex)
        if env_map is not None:
            c.env_map = int(env_map) + 1
        if env_transform is not None:
            c.env_transform = (C.c_float * 9)(*[float(x) for x in np.asarray(env_transform, np.float32).reshape(9)])
        self.rows = None
        if rows is not None:
            self.rows = (int(rows[0]), int(rows[1]))
            c.row_begin, c.row_end = self.rows
        self.c = c


class MeshShading:
    """crt_mesh_shading: per-triangle shading arrays of FLAT / SMOOTH triangles (any may be None)."""

    def __init__(self, flat_normal=None, vertex_normals=None, vertex_tangents=None, normal_mat_index=None, normal_mats=None, vertex_uv=None):
        f32 = lambda x: None if x is None else np.ascontiguousarray(x, np.float32)      # noqa: E731
        self.flat_normal, self.vertex_normals, self.vertex_tangents = f32(flat_normal), f32(vertex_normals), f32(vertex_tangents)
        self.normal_mat_index = None if normal_mat_index is None else np.ascontiguousarray(normal_mat_index, np.int32)
        self.normal_mats = f32(normal_mats)
        self.vertex_uv = f32(vertex_uv)                                       # [T, 6] or None
        c = CrtMeshShading()
        c.flat_normal, c.vertex_normals, c.vertex_tangents = _ptr(self.flat_normal), _ptr(self.vertex_normals), _ptr(self.vertex_tangents)
        c.normal_mat_index, c.normal_mats = _ptr(self.normal_mat_index), _ptr(self.normal_mats)
        c.num_normal_mats = 0 if self.normal_mats is None else len(self.normal_mats.reshape(-1, 9))
        c.vertex_uv = _ptr(self.vertex_uv)
        self.c = c


class CrtSceneFileView(C.Structure):
    _fields_ = [("input", CrtSceneInput), ("obj_material", C.POINTER(C.c_int32)), ("materials", C.c_void_p),
                ("material_names", C.POINTER(C.c_char_p)), ("num_materials", C.c_int32), ("has_mesh_shading", C.c_int32),
                ("mesh_shading", CrtMeshShading), ("has_film", C.c_int32), ("window_width", C.c_int32), ("window_height", C.c_int32),
                ("image_width", C.c_int32), ("image_height", C.c_int32), ("fov", C.c_float), ("has_camera", C.c_int32),
                ("camera_pos", C.c_float * 3), ("look_point", C.c_float * 3), ("has_env_light", C.c_int32),
                ("env_color", C.c_float * 3), ("env_intensity", C.c_float), ("num_skipped", C.c_int32), ("num_unsupported", C.c_int32),
                ("num_textures", C.c_int32), ("texture_names", C.POINTER(C.c_char_p)), ("textures", C.POINTER(CrtTexture)),
                ("material_diffuse_tex", C.POINTER(C.c_int32)), ("material_normal_tex", C.POINTER(C.c_int32)),
                ("env_type", C.c_int32), ("env_map_tex", C.c_int32), ("env_transform", C.c_float * 9), ("num_external", C.c_int32)]


class SceneFile:
    """crt_scene_file_load: a scene description file (scene/scenefile.h grammar, and the older one of the files the
    reference ships) as SceneInput + materials + camera / film / env light.  Arrays are copied out of the C object."""

    def __init__(self, path, L=None):
        L = L or lib()
        L.crt_scene_file_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.crt_scene_file_get.argtypes = [C.c_void_p]
        L.crt_scene_file_get.restype = C.POINTER(CrtSceneFileView)
        L.crt_scene_file_free.argtypes = [C.c_void_p]
        L.crt_scene_file_free.restype = None
        h = C.c_void_p()
        _check(L.crt_scene_file_load(os.fsencode(path), C.byref(h)), "crt_scene_file_load")
        try:
            v = L.crt_scene_file_get(h).contents
            n, nt, ns, nm = v.input.num_obj, v.input.num_tris, v.input.num_shapes, v.num_materials

            def arr(ptr, count, dtype):
                if not ptr or count == 0:
                    return np.zeros(count, dtype)
                return np.frombuffer(C.string_at(ptr, count * np.dtype(dtype).itemsize), dtype=dtype).copy()
            self.scene_input = SceneInput(arr(v.input.tri_verts, nt * 9, np.float32).reshape(-1, 9), arr(v.input.shapes, ns, SHAPE_DTYPE),
                                          arr(v.input.obj_type, n, np.int32), arr(v.input.obj_ref, n, np.int32))
            self.obj_material = arr(C.cast(v.obj_material, C.c_void_p).value, n, np.int32)
            self.materials = arr(v.materials, nm, MATERIAL_DTYPE)
            self.material_names = [v.material_names[i].decode() for i in range(nm)]
            self.mesh_shading = None
            if v.has_mesh_shading:
                ms = v.mesh_shading
                self.mesh_shading = MeshShading(arr(ms.flat_normal, nt * 3, np.float32).reshape(-1, 3),
                                                arr(ms.vertex_normals, nt * 9, np.float32).reshape(-1, 9),
                                                arr(ms.vertex_tangents, nt * 9, np.float32).reshape(-1, 9),
                                                arr(ms.normal_mat_index, nt, np.int32),
                                                arr(ms.normal_mats, ms.num_normal_mats * 9, np.float32).reshape(-1, 9),
                                                arr(ms.vertex_uv, nt * 6, np.float32).reshape(-1, 6) if ms.vertex_uv else None)
            self.film = (v.window_width, v.window_height, v.image_width, v.image_height, float(v.fov)) if v.has_film else None
            self.camera = (tuple(v.camera_pos), tuple(v.look_point)) if v.has_camera else None
            self.env_light = (tuple(v.env_color), float(v.env_intensity)) if v.has_env_light else None
            self.num_skipped, self.num_unsupported, self.num_external = v.num_skipped, v.num_unsupported, v.num_external
            # textures: names in order of first mention; texels where the loader could read them (PPM), else None
            self.texture_names = [v.texture_names[i].decode() for i in range(v.num_textures)]
            self.textures = []
            for i in range(v.num_textures):
                t = v.textures[i]
                if t.data:
                    dt = np.float32 if t.is_float else np.uint8
                    self.textures.append(arr(t.data, t.width * t.height * t.comp, dt).reshape(t.height, t.width, t.comp))
                else:
                    self.textures.append(None)
            self.material_diffuse_tex = arr(C.cast(v.material_diffuse_tex, C.c_void_p).value, nm, np.int32)
            self.material_normal_tex = arr(C.cast(v.material_normal_tex, C.c_void_p).value, nm, np.int32)
            self.env_type, self.env_map_tex = v.env_type, v.env_map_tex
            self.env_transform = np.array(list(v.env_transform), np.float32).reshape(3, 3)
        finally:
            L.crt_scene_file_free(h)

    def render_desc(self, max_depth, built=None, seed=0, samples_in_flight=0, world_radius=None, texels=None, rows=None):
        """RenderDesc for this scene.  The env light's world radius comes from the built accelerator as in
        preprocessLights (crt_env_world_radius) unless given.  texels: {file name: [h, w, 3] array} for textures the loader
        named but did not read (the .exr environment map: float32)."""
        env = None
        if self.env_light is not None:
            if world_radius is None:
                world_radius = env_world_radius(built)
            env = (self.env_light[0], self.env_light[1], world_radius)
        textures, diffuse, env_map, env_xf = None, None, None, None
        if self.texture_names:
            textures = []
            for name, t in zip(self.texture_names, self.textures):
                if texels and name in texels:
                    t = texels[name]
                if t is None:
                    raise ValueError("texture %r: the loader could not read it and no texels were given" % name)
                textures.append(t)
            diffuse = self.material_diffuse_tex
            if self.env_type == 1 and self.env_map_tex >= 0:
                env_map, env_xf = self.env_map_tex, self.env_transform
        return RenderDesc(self.scene_input, self.obj_material, self.materials, max_depth, seed=seed, env=env,
                          samples_in_flight=samples_in_flight, mesh_shading=self.mesh_shading, textures=textures,
                          material_diffuse_tex=diffuse, env_map=env_map, env_transform=env_xf, rows=rows)


def env_world_radius(built, L=None):
    """crt_env_world_radius (preprocessLights, buildscene.h:839-873) of a BuiltScene."""
    L = L or lib()
    L.crt_env_world_radius.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    r = C.c_float()
    _check(L.crt_env_world_radius(C.addressof(built.desc), C.byref(r)), "crt_env_world_radius")
    return float(r.value)


def bake_mesh(positions, indices, scaling, orientation, location, normals=None, tangents=None, L=None):
    """crt_bake_mesh (generateMeshTriangles): -> dict(tri_verts [T, 9], flat_normal [T, 3], vertex_normals / vertex_tangents
    [T, 9] or None, normal_mat [9])."""
    L = L or lib()
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.int32).reshape(-1)
    nrm = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    tan = None if tangents is None else np.ascontiguousarray(tangents, np.float32).reshape(-1, 3)
    m = CrtMesh()
    m.positions, m.normals, m.tangents, m.indices = _ptr(pos), _ptr(nrm), _ptr(tan), _ptr(idx)
    m.num_vertices, m.num_indices = len(pos), len(idx)
    e = CrtMeshEntry()
    e.scaling, e.orientation, e.location = _f3(scaling), _f3(orientation), _f3(location)
    T = len(idx) // 3
    out = {"tri_verts": np.zeros((T, 9), np.float32), "flat_normal": np.zeros((T, 3), np.float32),
           "vertex_normals": None if nrm is None else np.zeros((T, 9), np.float32),
           "vertex_tangents": None if tan is None else np.zeros((T, 9), np.float32), "normal_mat": np.zeros(9, np.float32)}
    _check(L.crt_bake_mesh(C.byref(m), C.byref(e), out["tri_verts"].ctypes.data, out["flat_normal"].ctypes.data,
                           _ptr(out["vertex_normals"]), _ptr(out["vertex_tangents"]), out["normal_mat"].ctypes.data), "crt_bake_mesh")
    return out


def obj_load(path):
    """crt_obj_load (loadOBJ): list of dicts with positions [V, 3], normals [Vn, 3], texcoords [Vt, 2], indices [T, 3], mat_name."""
    shapes, n = C.POINTER(CrtObjShape)(), C.c_int32()
    _check(lib().crt_obj_load(os.fsencode(path), C.byref(shapes), C.byref(n)), "crt_obj_load")
    out = []
    try:
        for i in range(n.value):
            s = shapes[i]
            arr = lambda p, cnt, dt: np.ctypeslib.as_array(p, (cnt,)).astype(dt).copy() if cnt else np.zeros(0, dt)   # noqa: E731
            out.append({"positions": arr(s.positions, s.num_positions, np.float32).reshape(-1, 3),
                        "normals": arr(s.normals, s.num_normals, np.float32).reshape(-1, 3),
                        "texcoords": arr(s.texcoords, s.num_texcoords, np.float32).reshape(-1, 2),
                        "indices": arr(s.indices, s.num_indices, np.int32).reshape(-1, 3), "mat_name": s.mat_name.decode()})
    finally:
        lib().crt_obj_free(shapes, n)
    return out


def mesh_normals_tangents(positions, indices, normals=None):
    """crt_mesh_normals_tangents: (normals [V, 3] -- given or generated as calcTriangleNormals does --, tangents [V, 3])."""
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.int32).reshape(-1)
    nin = None if normals is None or len(normals) == 0 else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    nout, tout = np.zeros_like(pos), np.zeros_like(pos)
    _check(lib().crt_mesh_normals_tangents(_ptr(pos), len(pos), _ptr(idx), len(idx), _ptr(nin), nout.ctypes.data, tout.ctypes.data),
           "crt_mesh_normals_tangents")
    return nout, tout


def material(type_, color, intensity=0.0, sigma=0.0, ior_in=1.0, ior_out=1.0, ks=(0.0, 0.0, 0.0), k=(0.0, 0.0, 0.0),
             uroughness=0.0, vroughness=None):
    """crt_material.  PLASTIC: color = kd, ks, uroughness = roughness.  GLASS: ior_in, ior_out, u/vroughness.
    METAL: color = eta, k, u/vroughness (METALS holds the reference's table)."""
    m = np.zeros(1, MATERIAL_DTYPE)[0]
    m["type"] = type_
    m["color"] = np.asarray(color, np.float32)
    m["intensity"] = intensity
    m["sigma"] = sigma
    m["ior_in"] = ior_in
    m["ior_out"] = ior_out
    m["ks"] = np.asarray(ks, np.float32)
    m["k"] = np.asarray(k, np.float32)
    m["uroughness"] = uroughness
    m["vroughness"] = uroughness if vroughness is None else vroughness
    return m


class Renderer:
    """crt_renderer: wavefront path tracing over a Scene (restricted form)."""

    def __init__(self, scene, desc, cam, film):
        self._scene, self._desc = scene, desc             # the C side keeps a pointer to the scene
        self.width, self.height = film.frame_res_width, film.frame_res_height
        if getattr(desc, "rows", None):                   # a band of the frame: the colour buffer holds these rows only
            self.height = desc.rows[1] - desc.rows[0]
        self._h = C.c_void_p()
        _check(lib().crt_renderer_create(scene._h, C.byref(desc.c), C.byref(cam), C.byref(film), C.byref(self._h)),
               "crt_renderer_create")

    def render(self, num_samples):
        _check(lib().crt_renderer_render(self._h, int(num_samples)), "crt_renderer_render")

    def read(self):
        """(color_buffer [h, w, 3] = running sum of samples, num_samples)"""
        buf = np.zeros((self.height, self.width, 3), np.float32)
        n = C.c_uint32()
        _check(lib().crt_renderer_read(self._h, buf.ctypes.data, C.byref(n)), "crt_renderer_read")
        return buf, n.value

    def reset(self, next_sample=0):
        """Clear the colour buffer; the next render() starts at sample index next_sample."""
        _check(lib().crt_renderer_reset(self._h, int(next_sample)), "crt_renderer_reset")

    def read_dev(self, d_color_buffer, stream=None):
        """Copy the colour buffer into device memory at d_color_buffer ([h*w*3] floats) on `stream`."""
        _check(lib().crt_renderer_read_dev(self._h, d_color_buffer, stream), "crt_renderer_read_dev")

    def stats(self):
        st = CrtRenderStats()
        _check(lib().crt_renderer_get_stats(self._h, C.byref(st)), "crt_renderer_get_stats")
        return {k: getattr(st, k) for k, _ in CrtRenderStats._fields_}

    def close(self):
        if self._h:
            lib().crt_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fill_shade_rec(scene_input, rays, hits, mesh_shading=None, obj_material=None, conventions=False, out=None, L=None):
    """crt_fill_shade_rec: the ShadeRec each hit leaves behind (host code).  Returns SHADE_REC_DTYPE[n]; fields the
    reference's intersector does not write keep the value `out` held (default: NaN, so that they show)."""
    L = L or lib()
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    hits = np.ascontiguousarray(hits, HIT_DTYPE)
    if out is None:
        out = np.zeros(len(rays), SHADE_REC_DTYPE)
        for f in ("hit_point", "normal", "dpdu", "dpdv", "wo", "uv"):
            out[f] = np.nan
    om = None if obj_material is None else np.ascontiguousarray(obj_material, np.int32)
    cin = scene_input.c_struct()
    _check(L.crt_fill_shade_rec(C.byref(cin), C.addressof(mesh_shading.c) if mesh_shading is not None else None, _ptr(om),
                                _ptr(rays), _ptr(hits), len(rays), 1 if conventions else 0, _ptr(out)), "crt_fill_shade_rec")
    return out


def image_state_save(path, color_buffer, num_samples):
    """saveImageState: color_buffer [h, w, 3] float32 sums."""
    cb = np.ascontiguousarray(color_buffer, np.float32)
    _check(lib().crt_image_state_save(os.fsencode(path), cb.ctypes.data, int(num_samples), cb.shape[1], cb.shape[0]),
           "crt_image_state_save")


def image_state_load(path):
    """readImageState -> (color_buffer [h, w, 3], num_samples)."""
    p, ns, w, h = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
    _check(lib().crt_image_state_load(os.fsencode(path), C.byref(p), C.byref(ns), C.byref(w), C.byref(h)), "crt_image_state_load")
    try:
        buf = np.ctypeslib.as_array((C.c_float * (w.value * h.value * 3)).from_address(p.value)).copy() if w.value * h.value \
            else np.zeros(0, np.float32)
    finally:
        lib().crt_image_state_free(p)
    return buf.reshape(h.value, w.value, 3), ns.value


def image_from_color_buffer(color_buffer, num_samples):
    """genImageFromColorBuffer + toneMap -> uint8 [h, w, 3]."""
    cb = np.ascontiguousarray(color_buffer, np.float32)
    img = np.zeros(cb.shape, np.uint8)
    _check(lib().crt_image_from_color_buffer(img.ctypes.data, cb.ctypes.data, cb.size // 3, int(num_samples)),
           "crt_image_from_color_buffer")
    return img


def texture_load_ppm(path):
    """crt_texture_load_ppm (PPM_read, imagefile.h:59-83): a binary P6 file as a [h, w, 3] uint8 array."""
    t = CrtTexture()
    _check(lib().crt_texture_load_ppm(os.fsencode(path), C.byref(t)), "crt_texture_load_ppm")
    try:
        return np.frombuffer(C.string_at(t.data, t.width * t.height * t.comp), np.uint8).reshape(t.height, t.width, t.comp).copy()
    finally:
        lib().crt_texture_free(C.byref(t))


def ppm_write(path, image):
    img = np.ascontiguousarray(image, np.uint8)
    _check(lib().crt_ppm_write(os.fsencode(path), img.ctypes.data, img.shape[1], img.shape[0]), "crt_ppm_write")


def _view(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    dt = np.dtype(dtype)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n)


class BuiltScene:
    """mvNonGridObjToStart + buildSceneAccel (scenedata.h:368, buildscene.h:698) on the host."""

    def __init__(self, scene_input, accel, sah=False):
        """sah=True: CRT_BUILD_SAH, the QBVH's split planes chosen by a surface-area heuristic (crt_build_scene_ex)."""
        self._h = C.c_void_p()
        self._input = scene_input            # keep arrays alive during the call only; build copies them
        cin = scene_input.c_struct()
        _check(lib().crt_build_scene_ex(C.byref(cin), int(accel), BUILD_SAH if sah else 0, C.byref(self._h)), "crt_build_scene_ex")
        self.desc = lib().crt_built_scene_desc(self._h).contents
        self.accel = int(accel)

    # numpy views into the built arrays (valid while this object lives)
    @property
    def obj_id(self):
        return _view(self.desc.obj_id, self.desc.num_obj, np.int32)

    @property
    def obj_type(self):
        return _view(self.desc.obj_type, self.desc.num_obj, np.int32)

    @property
    def obj_ref(self):
        return _view(self.desc.obj_ref, self.desc.num_obj, np.int32)

    @property
    def nodes4(self):
        return _view(self.desc.nodes4, self.desc.num_nodes4, NODE4_DTYPE)

    @property
    def nodes2(self):
        return _view(self.desc.nodes2, self.desc.num_nodes2, NODE2_DTYPE)

    @property
    def cell_start(self):
        return _view(self.desc.cell_start, self.desc.grid.num_cells + 1 if self.accel == ACCEL_GRID else 0, np.int32)

    @property
    def cell_items(self):
        cs = self.cell_start
        return _view(self.desc.cell_items, int(cs[-1]) if len(cs) else 0, np.int32)

    def close(self):
        if self._h:
            lib().crt_built_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """A scene resident in the HBM of one GPU; answers batches of rays."""

    def __init__(self, built, device=0):
        self._h = C.c_void_p()
        self.device = device
        _check(lib().crt_scene_create(int(device), C.byref(built.desc), C.byref(self._h)), "crt_scene_create")

    def set_traverse_mode(self, mode):
        _check(lib().crt_scene_set_traverse_mode(self._h, int(mode)), "crt_scene_set_traverse_mode")

    def set_ray_binning(self, on=True):
        _check(lib().crt_scene_set_ray_binning(self._h, 1 if on else 0), "crt_scene_set_ray_binning")

    def ray_binning(self):
        on = C.c_int()
        _check(lib().crt_scene_get_ray_binning(self._h, C.byref(on)), "crt_scene_get_ray_binning")
        return bool(on.value)

    def set_wide_nodes(self, on=True):
        """Closest-hit launches walk the 8-wide copy of the QBVH (default) or the 4-wide copies only; same hit records."""
        _check(lib().crt_scene_set_wide_nodes(self._h, 1 if on else 0), "crt_scene_set_wide_nodes")

    def set_node_compression(self, on=True):
        _check(lib().crt_scene_set_node_compression(self._h, 1 if on else 0), "crt_scene_set_node_compression")

    def traverse_mode(self):
        m = C.c_int()
        _check(lib().crt_scene_get_traverse_mode(self._h, C.byref(m)), "crt_scene_get_traverse_mode")
        return m.value

    def intersect(self, rays, out=None):
        """n x intersectTest.  rays: float32 [n, 6] (origin, direction).  Returns HIT_DTYPE[n]
        (written into `out` when given, e.g. a HostBuffer array)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        hits = np.empty(len(rays), HIT_DTYPE) if out is None else out
        assert hits.dtype == HIT_DTYPE and len(hits) == len(rays) and hits.flags.c_contiguous
        _check(lib().crt_intersect_batch(self._h, _ptr(rays), len(rays), _ptr(hits)), "crt_intersect_batch")
        return hits

    def shadow(self, rays, light_dist):
        """n x shadowIntersectTest; returns (t float32[n], occluded bool[n])."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        ld = np.ascontiguousarray(light_dist, dtype=np.float32).reshape(-1)
        assert len(ld) == len(rays)
        t = np.empty(len(rays), np.float32)
        occ = np.empty(len(rays), np.uint8)
        _check(lib().crt_shadow_batch(self._h, _ptr(rays), _ptr(ld), len(rays), _ptr(t), _ptr(occ)), "crt_shadow_batch")
        return t, occ.astype(bool)

    def intersect_dev(self, d_rays, n, d_hits, stream=None):
        """Device pointers (ints, e.g. torch.Tensor.data_ptr()); asynchronous on `stream`."""
        _check(lib().crt_intersect_batch_dev(self._h, d_rays, n, d_hits, stream), "crt_intersect_batch_dev")

    def intersect_dev_noalloc(self, d_rays, n, d_hits, stream=None):
        """crt_intersect_batch_dev_noalloc: the same on the library's file-scope scratch array (n <= NOALLOC_MAX_RAYS)."""
        _check(lib().crt_intersect_batch_dev_noalloc(self._h, d_rays, n, d_hits, stream), "crt_intersect_batch_dev_noalloc")

    def intersect_dev_slice(self, d_rays, n, d_hits, slice_index, num_slices, stream=None):
        """crt_intersect_batch_dev_slice: one share (a run of whole binning cells) of a batch several GPUs trace together."""
        _check(lib().crt_intersect_batch_dev_slice(self._h, d_rays, n, d_hits, int(slice_index), int(num_slices), stream),
               "crt_intersect_batch_dev_slice")

    def shadow_dev_slice(self, d_rays, d_light_dist, n, d_t, d_occ, slice_index, num_slices, stream=None):
        _check(lib().crt_shadow_batch_dev_slice(self._h, d_rays, d_light_dist, n, d_t, d_occ, int(slice_index), int(num_slices), stream),
               "crt_shadow_batch_dev_slice")

    def reserve_rays(self, max_rays):
        _check(lib().crt_scene_reserve_rays(self._h, int(max_rays)), "crt_scene_reserve_rays")

    def shadow_dev(self, d_rays, d_light_dist, n, d_t, d_occ, stream=None):
        _check(lib().crt_shadow_batch_dev(self._h, d_rays, d_light_dist, n, d_t, d_occ, stream), "crt_shadow_batch_dev")

    def intersect_dev_counted(self, d_rays, d_n, n_max, d_hits, stream=None):
        """crt_intersect_batch_dev_counted: the ray count is the device word at d_n (clamped to n_max)."""
        _check(lib().crt_intersect_batch_dev_counted(self._h, d_rays, d_n, n_max, d_hits, stream), "crt_intersect_batch_dev_counted")

    def shadow_dev_counted(self, d_rays, d_light_dist, d_n, n_max, d_t, d_occ, stream=None):
        _check(lib().crt_shadow_batch_dev_counted(self._h, d_rays, d_light_dist, d_n, n_max, d_t, d_occ, stream), "crt_shadow_batch_dev_counted")

    def enable_counters(self, on=True):
        _check(lib().crt_scene_enable_counters(self._h, 1 if on else 0), "crt_scene_enable_counters")

    def counters(self):
        c = CrtCounters()
        _check(lib().crt_scene_get_counters(self._h, C.byref(c)), "crt_scene_get_counters")
        return {"rays_closest": c.rays_closest, "rays_shadow": c.rays_shadow,
                "nodes_visited": c.nodes_visited, "prims_tested": c.prims_tested}

    def reset_counters(self):
        _check(lib().crt_scene_reset_counters(self._h), "crt_scene_reset_counters")

    def last_binning_ms(self):
        m = C.c_float()
        _check(lib().crt_scene_last_binning_ms(self._h, C.byref(m)), "crt_scene_last_binning_ms")
        return m.value

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(lib().crt_scene_last_kernel_ms(self._h, C.byref(ms)), "crt_scene_last_kernel_ms")
        return ms.value

    def close(self):
        if self._h:
            lib().crt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    n = C.c_int()
    _check(lib().crt_device_count(C.byref(n)), "crt_device_count")
    return n.value
