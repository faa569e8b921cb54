"""Synthetic scenes and ray sets for the traversal path (numpy, host side).

The reference ships no meshes (Buddha / San Miguel / Car / Coffee are absent) and
nothing can be fetched, so the large configurations run on procedural stand-ins of the stated
triangle counts.  Scene S1 takes its geometry verbatim from /root/reference/cornell_box.txt:70-246
(9 rectangles + 20 triangles; data, not code).
"""
import math

import numpy as np

from . import api

F32 = np.float32

# ---------------------------------------------------------------------------------------
# S1: cornell box (cornell_box.txt:70-246).  RECT rows: point, width, height.  TRI rows: v0 v1 v2.
# Floor and ceiling are listed twice in the file (:82-92 vs :112-122) -> exact-t tie class T1.
CORNELL_RECTS = [
    [213.0, 548.0, -227.0, 130.0, 0.0, 0.0, 0.0, 0.0, -105.0],
    [343.0, 548.0, -227.0, -130.0, 0.0, 0.0, 0.0, 0.0, -105.0],
    [0.0, 0.0, 0.0, 552.8, 0.0, 0.0, 0.0, 0.0, -559.2],
    [0.0, 548.8, 0.0, 0.0, 0.0, -559.2, 556.0, 0.0, 0.0],
    [0.0, 0.0, -559.2, 553.6, 0.0, 0.0, 0.0, 548.8, 0.0],
    [0.0, 0.0, 0.0, 0.0, 0.0, -559.2, 0.0, 548.8, 0.0],
    [552.8, 0.0, 0.0, 0.0, 548.8, 0.0, 0.0, 0.0, -559.2],
    [0.0, 0.0, 0.0, 552.8, 0.0, 0.0, 0.0, 0.0, -559.2],
    [0.0, 548.8, 0.0, 0.0, 0.0, -559.2, 556.0, 0.0, 0.0],
]
CORNELL_TRIS = [
    [423.0, 330.0, -247.0, 265.0, 330.0, -296.0, 314.0, 330.0, -456.0],
    [423.0, 330.0, -247.0, 314.0, 330.0, -456.0, 472.0, 330.0, -406.0],
    [423.0, 0.0, -247.0, 423.0, 330.0, -247.0, 472.0, 330.0, -406.0],
    [423.0, 0.0, -247.0, 472.0, 330.0, -406.0, 472.0, 0.0, -406.0],
    [472.0, 0.0, -406.0, 472.0, 330.0, -406.0, 314.0, 330.0, -456.0],
    [472.0, 0.0, -406.0, 314.0, 330.0, -456.0, 314.0, 0.0, -456.0],
    [314.0, 0.0, -456.0, 314.0, 330.0, -456.0, 265.0, 330.0, -296.0],
    [314.0, 0.0, -456.0, 265.0, 330.0, -296.0, 265.0, 0.0, -296.0],
    [265.0, 0.0, -296.0, 265.0, 330.0, -296.0, 423.0, 330.0, -247.0],
    [265.0, 0.0, -296.0, 423.0, 330.0, -247.0, 423.0, 0.0, -247.0],
    [130.0, 165.0, -65.0, 82.0, 165.0, -225.0, 240.0, 165.0, -272.0],
    [130.0, 165.0, -65.0, 240.0, 165.0, -272.0, 290.0, 165.0, -114.0],
    [290.0, 0.0, -114.0, 290.0, 165.0, -114.0, 240.0, 165.0, -272.0],
    [290.0, 0.0, -114.0, 240.0, 165.0, -272.0, 240.0, 0.0, -272.0],
    [130.0, 0.0, -65.0, 130.0, 165.0, -65.0, 290.0, 165.0, -114.0],
    [130.0, 0.0, -65.0, 290.0, 165.0, -114.0, 290.0, 0.0, -114.0],
    [82.0, 0.0, -225.0, 82.0, 165.0, -225.0, 130.0, 165.0, -65.0],
    [82.0, 0.0, -225.0, 130.0, 165.0, -65.0, 130.0, 0.0, -65.0],
    [240.0, 0.0, -272.0, 240.0, 165.0, -272.0, 82.0, 165.0, -225.0],
    [240.0, 0.0, -272.0, 82.0, 165.0, -225.0, 82.0, 0.0, -225.0],
]
CORNELL_CAMERA = ((278.0, 273.0, 800.0), (278.0, 273.0, 0.0))     # cornell_box.txt:2-3


def _normalize32(v):
    """vec3_normalize (util/vec.h:206-215) in float32."""
    v = np.asarray(v, F32)
    length = F32(np.sqrt(F32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
    if length == 0:
        return np.zeros(3, F32)
    return (v * F32(F32(1.0) / length)).astype(F32)


def _cross32(a, b):
    a = np.asarray(a, F32)
    b = np.asarray(b, F32)
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F32)


class ShapeList:
    """Accumulates analytic shapes + objects in insertion order."""

    def __init__(self):
        self.shapes = []
        self.obj_type = []
        self.obj_ref = []
        self.tris = []

    def _shape(self, type_, p=(), inner=-1, first_part=0, num_parts=0, inv_xform=None):
        rec = np.zeros(1, api.SHAPE_DTYPE)[0]
        rec["type"] = type_
        rec["inner"] = inner
        rec["first_part"] = first_part
        rec["num_parts"] = num_parts
        pp = np.zeros(12, F32)
        pp[:len(p)] = np.asarray(p, F32)
        rec["p"] = pp
        rec["inv_xform"] = np.eye(4, dtype=F32).reshape(16) if inv_xform is None else np.asarray(inv_xform, F32).reshape(16)
        self.shapes.append(rec)
        return len(self.shapes) - 1

    def _obj(self, type_, ref):
        self.obj_type.append(type_)
        self.obj_ref.append(ref)
        return len(self.obj_type) - 1

    def triangle(self, v, type_=api.TRIANGLE):
        self.tris.append(np.asarray(v, F32).reshape(9))
        return self._obj(type_, len(self.tris) - 1)

    def sphere(self, center, radius, min_theta=0.0, max_theta=3.14159, phi=3.14159):
        return self._obj(api.SPHERE, self._shape(api.SPHERE, list(center) + [radius, min_theta, max_theta, phi]))

    def plane(self, point, normal):
        return self._obj(api.PLANE, self._shape(api.PLANE, list(point) + list(normal)))

    def rectangle(self, point, width, height):
        normal = _normalize32(_cross32(width, height))         # parseRectEntry, scenefile.h:432-433
        return self._obj(api.RECTANGLE, self._shape(api.RECTANGLE, list(point) + list(width) + list(height) + list(normal)))

    def disk(self, center, normal, radius):
        return self._obj(api.DISK, self._shape(api.DISK, list(center) + list(normal) + [radius]))

    def box(self, inv_xform, x_span, y_span, z_span, shadow=True):
        """initBox (shapes/box.cpp:4-20): AABox of half spans wrapped in an instance."""
        hx, hy, hz = F32(x_span) / F32(2), F32(y_span) / F32(2), F32(z_span) / F32(2)
        inner = self._shape(api.AABOX, [-hx, -hy, -hz, hx, hy, hz, 1.0 if shadow else 0.0])
        return self._obj(api.INSTANCED, self._shape(api.INSTANCED, inner=inner, inv_xform=inv_xform))

    def torus(self, inv_xform, swept_radius, tube_radius, phi=3.14159):
        inner = self._shape(api.GENERICTORUS, [swept_radius, tube_radius, phi])
        return self._obj(api.INSTANCED, self._shape(api.INSTANCED, inner=inner, inv_xform=inv_xform))

    def open_cylinder(self, inv_xform, phi=3.14159):
        """initOpenCylinder (shapes/cylinder.cpp:3-18): radius 1, half height 1."""
        inner = self._shape(api.GENERICOPENCYLINDER, [1.0, 1.0, phi])
        return self._obj(api.INSTANCED, self._shape(api.INSTANCED, inner=inner, inv_xform=inv_xform))

    def solid_cylinder(self, inv_xform, radius, half_height, phi=3.14159):
        """initSolidCylinder (shapes/cylinder.cpp:29-75): compound {top disk, bottom disk, tube}."""
        first = self._shape(api.DISK, [0.0, half_height, 0.0, 0.0, 1.0, 0.0, radius])
        self._shape(api.DISK, [0.0, -half_height, 0.0, 0.0, -1.0, 0.0, radius])
        self._shape(api.GENERICOPENCYLINDER, [half_height, radius, phi])
        comp = self._shape(api.COMPOUND, [-radius, -half_height, -radius, radius, half_height, radius],
                           first_part=first, num_parts=3)
        return self._obj(api.INSTANCED, self._shape(api.INSTANCED, inner=comp, inv_xform=inv_xform))

    def to_input(self):
        tri = np.array(self.tris, F32).reshape(-1, 9) if self.tris else np.zeros((0, 9), F32)
        shapes = np.array(self.shapes, api.SHAPE_DTYPE) if self.shapes else np.zeros(0, api.SHAPE_DTYPE)
        return api.SceneInput(tri, shapes, np.array(self.obj_type, np.int32), np.array(self.obj_ref, np.int32))


def inv_transform(location, scale, euler_deg):
    """inv_scale * inv_rotation * inv_translation as parseBoxEntry composes it (scenefile.h:497-503),
    in float32.  rotation = Ry * (Rx * Rz) (eulerAngToMat4, util/math.cpp:80-93)."""
    def rot(axis, deg):
        th = F32(deg) * F32(math.pi / 180.0)
        s, c = F32(math.sin(th)), F32(math.cos(th))
        m = np.eye(4, dtype=F32)
        if axis == "x":
            m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, s, -s, c
        elif axis == "y":
            m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
        else:
            m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, s, -s, c
        return m
    rz, rx, ry = rot("z", euler_deg[2]), rot("x", euler_deg[1]), rot("y", euler_deg[0])
    rotation = (ry @ (rx @ rz)).astype(F32)
    inv_rot = rotation.T.copy()
    inv_scale = np.eye(4, dtype=F32)
    for i in range(3):
        inv_scale[i, i] = F32(1.0) / F32(scale[i])
    inv_tr = np.eye(4, dtype=F32)
    inv_tr[:3, 3] = -np.asarray(location, F32)
    return ((inv_scale @ inv_rot).astype(F32) @ inv_tr).astype(F32)


def cornell29():
    """S1: 9 rectangles + 20 triangles, insertion order of cornell_box.txt."""
    sl = ShapeList()
    for r in CORNELL_RECTS:
        sl.rectangle(r[0:3], r[3:6], r[6:9])
    for t in CORNELL_TRIS:
        sl.triangle(t, api.TRIANGLE)
    return sl.to_input()


# cornell_box.txt:5-53 (DIFF_COLOR / COLOR / INTENSITY values; data) and the MATERIAL line of each object (:68-246)
CORNELL_MATERIALS = [
    ("floor", api.MAT_MATTE, (0.740064, 0.742313, 0.733934), 0.0),
    ("ceiling", api.MAT_MATTE, (0.740064, 0.742313, 0.733934), 0.0),
    ("backwall", api.MAT_MATTE, (0.740064, 0.742313, 0.733934), 0.0),
    ("leftwall", api.MAT_MATTE, (0.366046, 0.0371827, 0.0416385), 0.0),
    ("rightwall", api.MAT_MATTE, (0.162928, 0.408903, 0.0833759), 0.0),
    ("emissive1", api.MAT_EMISSIVE, (1.0, 0.85, 0.5), 90.0),
]
CORNELL_RECT_MATERIALS = ["emissive1", "emissive1", "floor", "ceiling", "backwall", "rightwall", "leftwall", "floor", "ceiling"]
CORNELL_TRI_MATERIAL = "ceiling"


def cornell29_render(max_depth=5, seed=1, samples_in_flight=0):
    """S1 with its materials and lights: (SceneInput, RenderDesc).  ENV_LIGHT CONSTANT WHITE, intensity 0
    (cornell_box.txt:57-60); world_radius is irrelevant at intensity 0 and set to the box diagonal x 2."""
    si = cornell29()
    names = [m[0] for m in CORNELL_MATERIALS]
    mats = np.array([api.material(m[1], m[2], m[3]) for m in CORNELL_MATERIALS], api.MATERIAL_DTYPE)
    obj_mat = [names.index(n) for n in CORNELL_RECT_MATERIALS] + [names.index(CORNELL_TRI_MATERIAL)] * len(CORNELL_TRIS)
    desc = api.RenderDesc(si, obj_mat, mats, max_depth, seed=seed, env=((1.0, 1.0, 1.0), 0.0, 1930.0),
                          samples_in_flight=samples_in_flight)
    return si, desc


def cornell_camera(width, height, fov=0.6911, L=None):
    """Camera of cornell_box.txt:2-3 with the reference's default focal distance; fov as config (main.cpp fov)."""
    cam = api.make_camera(CORNELL_CAMERA[0], CORNELL_CAMERA[1], L=L)
    return cam, api.make_film(cam, fov, width, height, L=L)


def render_room(n_lat=64, n_lon=128, seed=3, max_depth=5, rng_seed=7, env_intensity=0.25, samples_in_flight=0,
                tri_type=None, mirrors=False, glass=False, rough=False, glossy=False):
    """A displaced blob (TRIANGLE objects) in a unit room of 12 triangles, two matte spheres, a sphere
    light, a disk light, a rectangle light under the ceiling (slightly tilted, so its box is not flat)
    and a constant environment light seen through nothing (the room is closed) but sampled:
    every branch of estimateDirect.  Returns (SceneInput, RenderDesc, camera position, look point)."""
    tri_type = api.TRIANGLE if tri_type is None else tri_type
    blob = blob_mesh(n_lat, n_lon, seed, radius=1.0, displace=0.45) * F32(0.3) + np.tile(np.array([0.5, 0.42, 0.45], F32), 3)
    sl = ShapeList()
    mat_of = []
    for tri in room_tris((0, 0, 0), (1, 1, 1)):
        sl.triangle(tri, tri_type)
    walls = [0, 0, 1, 1, 0, 0, 0, 0, 2, 2, 0, 0]          # one red and one green wall pair, rest white
    if mirrors:
        walls[10] = walls[11] = 8                         # one wall is a mirror
    mat_of += walls
    for tri in blob.astype(F32):
        sl.triangle(tri, tri_type)
    mat_of += [3] * len(blob)
    sl.rectangle((0.35, 0.97, 0.35), (0.3, 0.02, 0.0), (0.0, 0.0, 0.3)); mat_of.append(4)
    sl.rectangle((0.35, 0.97, 0.65), (0.3, 0.02, 0.0), (0.0, 0.0, -0.3)); mat_of.append(4)   # same place, opposite normal
    sl.sphere((0.18, 0.12, 0.75), 0.12); mat_of.append(9 if glass else 5)
    if glass:                                             # a thin pane across the room in front of the camera
        sl.rectangle((0.1, 0.1, 0.8), (0.8, 0.0, 0.02), (0.0, 0.7, 0.0)); mat_of.append(9)
    sl.sphere((0.82, 0.10, 0.70), 0.10); mat_of.append(8 if (mirrors or glossy) else 0)
    sl.sphere((0.85, 0.80, 0.20), 0.05); mat_of.append(6)                                   # sphere light
    sl.disk((0.02, 0.5, 0.5), _normalize32((1.0, 0.1, 0.0)), 0.15); mat_of.append(7)            # disk light on the left wall
    mats = np.array([
        api.material(api.MAT_MATTE, (0.74, 0.74, 0.73)), api.material(api.MAT_MATTE, (0.37, 0.04, 0.04)),
        api.material(api.MAT_MATTE, (0.16, 0.41, 0.08)), api.material(api.MAT_MATTE, (0.55, 0.5, 0.3)),
        api.material(api.MAT_EMISSIVE, (1.0, 0.85, 0.5), 12.0), api.material(api.MAT_MATTE, (0.2, 0.3, 0.7)),
        api.material(api.MAT_EMISSIVE, (0.6, 0.8, 1.0), 40.0), api.material(api.MAT_EMISSIVE, (1.0, 0.4, 0.3), 6.0),
        api.material(api.MAT_MIRROR, (0.9, 0.9, 0.85)),
        api.material(api.MAT_TRANSPARENT, (1.0, 1.0, 1.0), ior_in=1.5, ior_out=1.0),
    ], api.MATERIAL_DTYPE)
    if rough:                                             # Oren-Nayar walls and blob (sigma in degrees)
        mats[0]["sigma"], mats[1]["sigma"], mats[3]["sigma"] = 20.0, 35.0, 60.0
    if glossy:                                            # PLASTIC walls and blob, a gold and a rough-glass sphere, a brushed-copper wall
        mats[0] = api.material(api.MAT_PLASTIC, (0.6, 0.6, 0.58), ks=(0.25, 0.25, 0.25), uroughness=0.12)
        mats[3] = api.material(api.MAT_PLASTIC, (0.5, 0.35, 0.1), ks=(0.4, 0.4, 0.3), uroughness=0.3)
        gold, copper = api.METALS["GOLD"], api.METALS["COPPER"]
        mats[5] = api.material(api.MAT_METAL, gold[0], k=gold[1], uroughness=0.08)
        mats[2] = api.material(api.MAT_METAL, copper[0], k=copper[1], uroughness=0.05, vroughness=0.35)     # anisotropic
        mats[8] = api.material(api.MAT_GLASS, (1, 1, 1), ior_in=1.5, ior_out=1.0, uroughness=0.1, vroughness=0.1)
    si = sl.to_input()
    env = ((1.0, 1.0, 1.0), env_intensity, 3.5) if env_intensity is not None else None
    desc = api.RenderDesc(si, mat_of, mats, max_depth, seed=rng_seed, env=env, samples_in_flight=samples_in_flight)
    return si, desc, (0.5, 0.5, 0.98), (0.5, 0.45, 0.0)


def cube_tris(center, half):
    c = np.asarray(center, F32)
    h = F32(half)
    v = np.array([[x, y, z] for x in (-h, h) for y in (-h, h) for z in (-h, h)], F32) + c
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, cc, d in quads:
        tris.append(np.concatenate([v[a], v[b], v[cc]]))
        tris.append(np.concatenate([v[a], v[cc], v[d]]))
    return np.array(tris, F32)


def analytic_scene(rotated=True):
    """S2: the shapes of example_scene.txt:32-78 (sphere, plane, rectangle, triangle, disk) plus an
    instanced box / torus(2.0, 0.1) / open cylinder / solid cylinder and a 12-triangle cube."""
    sl = ShapeList()
    sl.sphere((-2.0, 1.0, 0.0), 1.0)
    sl.plane((0.0, -1.0, 0.0), (0.0, 1.0, 0.0))
    sl.rectangle((1.0, -1.0, -3.0), (4.0, 0.0, 0.0), (0.0, 4.0, -4.0))
    sl.triangle([3.0, 0.0, 2.0, 5.0, 0.0, 2.0, 4.0, 2.0, 2.0])
    sl.disk((3.0, 3.0, 0.0), _normalize32((1.0, 0.0, 1.0)), 1.0)
    e = (20.0, 35.0, 10.0) if rotated else (0.0, 0.0, 0.0)
    s = (1.5, 0.75, 1.25) if rotated else (1.0, 1.0, 1.0)
    sl.box(inv_transform((-4.0, 0.5, -2.0), s, e), 1.0, 2.0, 1.5)
    sl.torus(inv_transform((0.0, 0.5, 3.0), (1.0, 1.0, 1.0) if not rotated else (1.0, 1.5, 1.0), e), 2.0, 0.1)
    sl.open_cylinder(inv_transform((-1.0, 0.0, -4.0), s, e))
    sl.solid_cylinder(inv_transform((5.0, 0.0, -1.0), s, e), 0.5, 1.0)
    for t in cube_tris((2.0, 0.0, -1.0), 0.5):
        sl.triangle(t, api.FLAT_TRIANGLE)
    return sl.to_input()


def analytic_render(max_depth=5, rng_seed=2, env_intensity=0.6, samples_in_flight=0, rotated=True, glossy=False):
    """BASELINE.json configs[1] as a render: the shapes of example_scene.txt:32-78 -- sphere, ground plane, rectangle,
    triangle, disk, torus(2.0, 0.1) -- plus an instanced box, open cylinder and solid cylinder and the 12-triangle
    cube, lit by a rectangle light overhead, a small sphere light and a constant sky ("direct light sampling").
    example_scene.txt itself names no light and its material grammar is the old one, so lights and materials are
    this repository's.  Returns (SceneInput, RenderDesc, camera position, look point)."""
    sl = ShapeList()
    mat_of = []
    sl.sphere((-2.0, 1.0, 0.0), 1.0); mat_of.append(5 if glossy else 1)
    sl.plane((0.0, -1.0, 0.0), (0.0, 1.0, 0.0)); mat_of.append(0)
    sl.rectangle((1.0, -1.0, -3.0), (4.0, 0.0, 0.0), (0.0, 4.0, -4.0)); mat_of.append(1)
    sl.triangle([3.0, 0.0, 2.0, 5.0, 0.0, 2.0, 4.0, 2.0, 2.0]); mat_of.append(1)
    sl.disk((3.0, 3.0, 0.0), _normalize32((1.0, 0.0, 1.0)), 1.0); mat_of.append(2)
    e = (20.0, 35.0, 10.0) if rotated else (0.0, 0.0, 0.0)
    s = (1.5, 0.75, 1.25) if rotated else (1.0, 1.0, 1.0)
    sl.box(inv_transform((-4.0, 0.5, -2.0), s, e), 1.0, 2.0, 1.5); mat_of.append(3)
    sl.torus(inv_transform((0.0, 0.5, 3.0), (1.0, 1.5, 1.0) if rotated else (1.0, 1.0, 1.0), e), 2.0, 0.1); mat_of.append(6 if glossy else 1)
    sl.open_cylinder(inv_transform((-1.0, 0.0, -4.0), s, e)); mat_of.append(2)
    sl.solid_cylinder(inv_transform((5.0, 0.0, -1.0), s, e), 0.5, 1.0); mat_of.append(3)
    for t in cube_tris((2.0, 0.0, -1.0), 0.5):
        sl.triangle(t, api.FLAT_TRIANGLE); mat_of.append(2)
    sl.rectangle((-1.5, 6.0, -1.5), (3.0, 0.0, 0.0), (0.0, 0.0, 3.0)); mat_of.append(4)          # light, facing down (normal = w x h = -y)
    sl.sphere((-5.0, 2.5, 2.0), 0.4); mat_of.append(7)                                          # sphere light
    gold = api.METALS["GOLD"]
    mats = np.array([
        api.material(api.MAT_MATTE, (0.55, 0.55, 0.55)), api.material(api.MAT_MATTE, (0.5, 0.5, 0.5), sigma=20.0 if glossy else 0.0),
        api.material(api.MAT_MATTE, (0.6, 0.55, 0.1)), api.material(api.MAT_MATTE, (0.2, 0.35, 0.6)),
        api.material(api.MAT_EMISSIVE, (1.0, 0.95, 0.85), 14.0),
        api.material(api.MAT_PLASTIC, (0.5, 0.2, 0.2), ks=(0.3, 0.3, 0.3), uroughness=0.15),
        api.material(api.MAT_METAL, gold[0], k=gold[1], uroughness=0.1),
        api.material(api.MAT_EMISSIVE, (0.6, 0.8, 1.0), 25.0),
    ], api.MATERIAL_DTYPE)
    si = sl.to_input()
    env = ((0.7, 0.8, 1.0), env_intensity, 60.0) if env_intensity is not None else None
    desc = api.RenderDesc(si, mat_of, mats, max_depth, seed=rng_seed, env=env, samples_in_flight=samples_in_flight)
    return si, desc, (0.0, 2.5, 9.0), (0.0, 0.5, 0.0)


def random_tris(n, seed=1, extent=1.0, size=0.05):
    """n small random triangles in [0, extent]^3 (the soup of SURVEY.md section 6)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.random((n, 1, 3), dtype=F32) * F32(extent)
    d = (rng.random((n, 3, 3), dtype=F32) - F32(0.5)) * F32(2.0 * size)
    return api.SceneInput((c + d).astype(F32).reshape(n, 9))


# ---------------------------------------------------------------------------------------
# procedural meshes for the large configurations

def _value_noise(p, seed, octaves=3):
    """cheap lattice value noise on points p [n,3] in float32, deterministic in seed."""
    out = np.zeros(len(p), F32)
    amp, freq = F32(1.0), F32(2.0)
    for o in range(octaves):
        q = p * freq
        i = np.floor(q).astype(np.int64)
        f = (q - i).astype(F32)
        f = f * f * (F32(3.0) - F32(2.0) * f)
        acc = np.zeros(len(p), F32)
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    h = ((i[:, 0] + dx) * 73856093) ^ ((i[:, 1] + dy) * 19349663) ^ ((i[:, 2] + dz) * 83492791) ^ (seed * 2654435761 + o)
                    h = (h * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFF
                    r = (h.astype(np.float64) / 4294967295.0).astype(F32)
                    w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                    acc += r * w
        out += amp * acc
        amp *= F32(0.5)
        freq *= F32(2.0)
    return out


def blob_mesh(n_lat, n_lon, seed, radius=1.0, displace=0.25):
    """A noise-displaced lat-long sphere: 2 * n_lat * n_lon triangles (degenerate-free: the poles
    are cut at a small cap angle).  Returns float32 [ntri, 9]."""
    th = np.linspace(0.02 * math.pi, 0.98 * math.pi, n_lat + 1)
    ph = np.linspace(0.0, 2.0 * math.pi, n_lon + 1)[:-1]
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).astype(F32)
    r = F32(radius) * (F32(1.0) + F32(displace) * (_value_noise(d.reshape(-1, 3) * F32(1.5) + F32(3.0), seed) - F32(0.9)))
    v = d * r.reshape(n_lat + 1, n_lon, 1)
    a = v[:-1, :, :]
    b = v[1:, :, :]
    a2 = np.roll(a, -1, axis=1)
    b2 = np.roll(b, -1, axis=1)
    t1 = np.concatenate([a, b, b2], -1).reshape(-1, 9)
    t2 = np.concatenate([a, b2, a2], -1).reshape(-1, 9)
    return np.concatenate([t1, t2], 0).astype(F32)


def _rot_matrix(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], F32)


def room_tris(lo, hi):
    """12 triangles: the six walls of an axis-aligned room."""
    lo = np.asarray(lo, F32)
    hi = np.asarray(hi, F32)
    c = (lo + hi) * F32(0.5)
    t = cube_tris((0, 0, 0), 1.0).reshape(-1, 3, 3)
    return (t * ((hi - lo) * F32(0.5)) + c).reshape(-1, 9).astype(F32)


def courtyard(n_copies, n_lat, n_lon, seed=5, room=10.0):
    """S5-style stand-in for "San Miguel, instanced": a box room plus n_copies rotated, scaled,
    translated copies of one displaced blob, FLATTENED to world space as generateMeshTriangles
    does for every OBJECT MESH entry (buildscene.h:241-311).  2*n_lat*n_lon*n_copies + 12 triangles."""
    rng = np.random.Generator(np.random.PCG64(seed))
    base = blob_mesh(n_lat, n_lon, seed).reshape(-1, 3)
    parts = [room_tris((0, 0, 0), (room, room * 0.5, room))]
    for k in range(n_copies):
        R = _rot_matrix(rng)
        s = F32(0.15 + 0.35 * rng.random())
        pos = np.array([rng.random() * room * 0.9 + room * 0.05, rng.random() * room * 0.4 + room * 0.05,
                        rng.random() * room * 0.9 + room * 0.05], F32)
        parts.append(((base @ R.T) * s + pos).astype(F32).reshape(-1, 9))
    return api.SceneInput(np.concatenate(parts, 0))


def icosphere(level, seed=3, displace=0.12):
    """Indexed icosphere subdivided `level` times (20 * 4^level well-shaped triangles, shared vertices), radius displaced
    by a few low-frequency sine lobes: the S3 "Buddha ~1M triangles" stand-in (level 8: 1 310 720 triangles).
    Returns (positions [V, 3], normals [V, 3], tangents [V, 3], indices [T, 3])."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
        e.sort(axis=1)
        uniq, inv = np.unique(e[:, 0] * (len(v) + 1) + e[:, 1], return_inverse=True)
        a, b = uniq // (len(v) + 1), uniq % (len(v) + 1)
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = len(v) + inv.reshape(3, -1)                            # midpoint index of edge k of every face
        v = np.concatenate([v, mid], 0)
        f = np.concatenate([np.stack([f[:, 0], m[0], m[2]], 1), np.stack([f[:, 1], m[1], m[0]], 1),
                            np.stack([f[:, 2], m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)], 0)
    rng = np.random.default_rng(seed)
    rad = np.ones(len(v))
    for k in range(6):
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        rad += displace / (k + 1) * np.sin((2 + k) * (v @ d) * math.pi + rng.random() * 6.28)
    pos = v * rad[:, None]
    nrm = v.copy()                                                 # the sphere's normals: smooth, close enough for a displaced blob
    tan = np.cross(np.array([0.0072, 1.0, 0.0034]), nrm)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    return pos.astype(F32), nrm.astype(F32), tan.astype(F32), f.astype(np.int32)


def smooth_blob_room(n_lat=24, n_lon=48, smooth=True, max_depth=5, rng_seed=3, seed=11, blob_material=None, mesh=None):
    """A unit room of TRIANGLE walls with a sphere light and, baked through crt_bake_mesh (generateMeshTriangles) with a
    non-uniform scale and a rotation, one indexed lat-long blob with per-vertex normals and tangents: SMOOTH_TRIANGLE
    objects (smooth=True) or FLAT_TRIANGLE objects carrying FlatTriangle.normal.  Returns (SceneInput, RenderDesc, pos, look)."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0.08, math.pi - 0.08, n_lat + 1, dtype=np.float64)
    ph = np.linspace(0.0, 2.0 * math.pi, n_lon, endpoint=False, dtype=np.float64)
    T, P = np.meshgrid(th, ph, indexing="ij")
    rad = 1.0 + 0.25 * np.sin(3 * T) * np.cos(4 * P) + 0.05 * rng.standard_normal(T.shape)
    pos = np.stack([rad * np.sin(T) * np.cos(P), rad * np.cos(T), rad * np.sin(T) * np.sin(P)], -1).reshape(-1, 3).astype(F32)
    nrm = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    tan = np.stack([-np.sin(P), np.zeros_like(P), np.cos(P)], -1).reshape(-1, 3).astype(F32)
    idx = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
            c, d = a + n_lon, b + n_lon
            idx += [a, c, b, b, c, d]
    if mesh is not None:                                           # (positions, normals, tangents, indices) instead of the lat-long blob
        pos, nrm, tan, idx = mesh
    baked = api.bake_mesh(pos, idx, scaling=(0.22, 0.3, 0.2), orientation=(25.0, -15.0, 40.0), location=(0.5, 0.4, 0.45),
                          normals=np.asarray(nrm, F32), tangents=tan)
    walls = room_tris((0, 0, 0), (1, 1, 1))
    nw, nb = len(walls), len(baked["tri_verts"])
    tris = np.concatenate([walls, baked["tri_verts"]], 0)
    sl = ShapeList()
    sl.sphere((0.8, 0.8, 0.3), 0.06)
    shapes = np.array(sl.shapes, api.SHAPE_DTYPE)
    obj_type = np.concatenate([np.full(nw, api.TRIANGLE), np.full(nb, api.SMOOTH_TRIANGLE if smooth else api.FLAT_TRIANGLE), [api.SPHERE]]).astype(np.int32)
    obj_ref = np.concatenate([np.arange(nw + nb), [0]]).astype(np.int32)
    si = api.SceneInput(tris, shapes, obj_type, obj_ref)
    pad3, pad9 = np.zeros((nw, 3), F32), np.zeros((nw, 9), F32)
    ms = api.MeshShading(flat_normal=np.concatenate([pad3, baked["flat_normal"]], 0),
                         vertex_normals=np.concatenate([pad9, baked["vertex_normals"]], 0),
                         vertex_tangents=np.concatenate([pad9, baked["vertex_tangents"]], 0),
                         normal_mat_index=np.zeros(nw + nb, np.int32), normal_mats=baked["normal_mat"].reshape(1, 9))
    mats = np.array([api.material(api.MAT_MATTE, (0.7, 0.7, 0.68)), api.material(api.MAT_MATTE, (0.6, 0.45, 0.25), sigma=25.0),
                     api.material(api.MAT_EMISSIVE, (1.0, 0.9, 0.8), 60.0)], api.MATERIAL_DTYPE)
    if blob_material is not None:
        mats[1] = blob_material
    mat_of = [0] * nw + [1] * nb + [2]
    desc = api.RenderDesc(si, mat_of, mats, max_depth, seed=rng_seed, env=((0.6, 0.7, 1.0), 0.2, 3.5), mesh_shading=ms)
    return si, desc, (0.5, 0.5, 0.98), (0.5, 0.42, 0.0)


def courtyard_render(n_copies, n_lat, n_lon, seed=5, room=10.0, max_depth=5, rng_seed=1, env_intensity=1.0,
                     samples_in_flight=0, glossy=False):
    """The courtyard() stand-in as a render scene: TRIANGLE objects, open to the sky (the two ceiling
    triangles removed), lit by a constant environment light -- "incoherent diffuse bounces" of
    BASELINE.json's last config.  White walls, four blob colours.  Returns (SceneInput, RenderDesc,
    camera position, look point)."""
    si0 = courtyard(n_copies, n_lat, n_lon, seed, room)
    tris = np.delete(si0.tri_verts, [6, 7], axis=0)                 # y = +h face of room_tris
    n = len(tris)
    si = api.SceneInput(tris, None, np.full(n, api.TRIANGLE, np.int32), np.arange(n, dtype=np.int32))
    per_copy = 2 * n_lat * n_lon
    mat_of = np.zeros(n, np.int32)
    mat_of[10:] = 1 + (np.arange(n - 10) // per_copy) % 4
    mats = np.array([api.material(api.MAT_MATTE, (0.74, 0.74, 0.73)), api.material(api.MAT_MATTE, (0.6, 0.3, 0.2)),
                     api.material(api.MAT_MATTE, (0.3, 0.55, 0.25)), api.material(api.MAT_MATTE, (0.25, 0.35, 0.6)),
                     api.material(api.MAT_MATTE, (0.7, 0.65, 0.4))], api.MATERIAL_DTYPE)
    if glossy:                                                      # the blobs in plastic, gold, rough glass and brushed copper
        gold, copper = api.METALS["GOLD"], api.METALS["COPPER"]
        mats[1] = api.material(api.MAT_PLASTIC, (0.6, 0.3, 0.2), ks=(0.3, 0.3, 0.3), uroughness=0.2)
        mats[2] = api.material(api.MAT_METAL, gold[0], k=gold[1], uroughness=0.1)
        mats[3] = api.material(api.MAT_GLASS, (1, 1, 1), ior_in=1.5, ior_out=1.0, uroughness=0.15, vroughness=0.15)
        mats[4] = api.material(api.MAT_METAL, copper[0], k=copper[1], uroughness=0.05, vroughness=0.3)
    desc = api.RenderDesc(si, mat_of, mats, max_depth, seed=rng_seed, env=((1.0, 1.0, 1.0), env_intensity, 4.0 * room),
                          samples_in_flight=samples_in_flight)
    return si, desc, (room * 0.5, room * 0.3, room * 0.97), (room * 0.5, room * 0.2, 0.0)


def buddha_like(n_lat=512, n_lon=1024, seed=3):
    """S3-style stand-in for the ~1M-triangle Buddha: one finely displaced blob in a room."""
    blob = blob_mesh(n_lat, n_lon, seed, radius=1.0, displace=0.45) * F32(0.4) + np.tile(np.array([0.5, 0.5, 0.5], F32), 3)
    return api.SceneInput(np.concatenate([room_tris((0, 0, 0), (1, 1, 1)), blob.astype(F32)], 0))


# ---------------------------------------------------------------------------------------
# ray sets

def lit_mesh_scene(n_lat=512, n_lon=1024, seed=3):
    """A ~1M-triangle displaced blob in a room with the usual non-mesh furniture of a render scene:
    an area light (rectangle) under the ceiling, a wall rectangle, two spheres, a disk, and a ground
    plane (not under the accelerator).  Every shape under the accelerator has a bounding box."""
    sl = ShapeList()
    sl.rectangle((0.35, 0.995, 0.35), (0.3, 0.0, 0.0), (0.0, 0.0, 0.3))            # light
    sl.sphere((0.2, 0.15, 0.8), 0.12)
    sl.plane((0.0, -0.01, 0.0), (0.0, 1.0, 0.0))
    sl.rectangle((0.02, 0.1, 0.1), (0.0, 0.5, 0.0), (0.0, 0.0, 0.6))                # panel near a wall
    sl.disk((0.8, 0.05, 0.2), _normalize32((0.0, 1.0, 0.2)), 0.1)
    sl.sphere((0.8, 0.8, 0.8), 0.1, 0.3, 2.5, 2.0)                                  # partial sphere
    base = sl.to_input()
    blob = blob_mesh(n_lat, n_lon, seed, radius=1.0, displace=0.45) * F32(0.4) + np.tile(np.array([0.5, 0.5, 0.5], F32), 3)
    tris = np.concatenate([room_tris((0, 0, 0), (1, 1, 1)), blob.astype(F32)], 0)
    n_t = len(tris)
    obj_type = np.concatenate([base.obj_type, np.full(n_t, api.SMOOTH_TRIANGLE, np.int32)])
    obj_ref = np.concatenate([base.obj_ref, np.arange(n_t, dtype=np.int32)])
    return api.SceneInput(tris, base.shapes, obj_type, obj_ref)


def scene_bounds(si):
    v = si.tri_verts.reshape(-1, 3)
    if len(v) == 0:
        return np.array([-6, -2, -6], F32), np.array([6, 4, 6], F32)
    return v.min(0), v.max(0)


def random_rays(n, lo, hi, seed=0xC0FFEE):
    """origins uniform in [lo, hi], directions uniform on the sphere (unit length, float32)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    o = (rng.random((n, 3), dtype=F32) * (np.asarray(hi, F32) - np.asarray(lo, F32)) + np.asarray(lo, F32)).astype(F32)
    d = rng.normal(size=(n, 3)).astype(F32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(F32)
    return np.concatenate([o, d.astype(F32)], 1).astype(F32)


def camera_rays(width, height, eye, look, fov_deg=40.0, seed=None):
    """pinhole rays through pixel centres (coherent; sanity only)."""
    eye = np.asarray(eye, np.float64)
    w = eye - np.asarray(look, np.float64)
    w /= np.linalg.norm(w)
    u = np.cross([0.0072, 1.0, 0.0034], w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    half = math.tan(math.radians(fov_deg) / 2)
    xs = ((np.arange(width) + 0.5) / width * 2 - 1) * half * width / height
    ys = ((np.arange(height) + 0.5) / height * 2 - 1) * half
    X, Y = np.meshgrid(xs, ys)
    d = X[..., None] * u + Y[..., None] * v - w
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(eye, d.shape)
    return np.concatenate([o, d], -1).reshape(-1, 6).astype(F32)


def axis_parallel_rays(n, lo, hi, seed=7):
    """rays with one or two exactly-zero (and some -0.0) direction components: classes T2/T3."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r = random_rays(n, lo, hi, seed)
    k = rng.integers(0, 3, n)
    r[np.arange(n), 3 + k] = 0.0
    neg = rng.random(n) < 0.25
    r[neg, 3 + k[neg]] = F32(-0.0)
    two = rng.random(n) < 0.2
    k2 = (k + 1) % 3
    r[two, 3 + k2[two]] = 0.0
    nrm = np.linalg.norm(r[:, 3:], axis=1, keepdims=True)
    nrm[nrm == 0] = 1
    r[:, 3:] = (r[:, 3:] / nrm).astype(F32)
    return r.astype(F32)


def tri_normals(tri_verts, ids):
    t = tri_verts[ids].reshape(-1, 3, 3).astype(np.float64)
    n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1
    return n / ln


def bounce_rays(trace, si, primary, bounces=3, seed=0xC0FFEE, keep_primary=False):
    """R-incoherent: trace `primary`, spawn a cosine-hemisphere direction about the
    geometric normal at every hit (origin = hit point, no offset: resetRay, trace.h:496), repeat;
    return the concatenated, shuffled bounce-1..bounces rays.  `trace(rays) -> HIT_DTYPE[n]` is
    the engine under test or the oracle; triangle-only scenes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = [primary] if keep_primary else []
    rays = primary
    for _ in range(bounces):
        hits = trace(rays)
        ok = hits["obj"] >= 0
        if not ok.any():
            break
        r = rays[ok]
        h = hits[ok]
        p = (r[:, :3].astype(np.float64) + r[:, 3:].astype(np.float64) * h["t"].astype(np.float64)[:, None])
        n = tri_normals(si.tri_verts, h["obj"])
        flip = (n * r[:, 3:]).sum(1) > 0
        n[flip] = -n[flip]
        u1, u2 = rng.random(len(r)), rng.random(len(r))
        rr, ph = np.sqrt(u1), 2 * math.pi * u2
        a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
        tx = np.cross(a, n)
        tx /= np.linalg.norm(tx, axis=1, keepdims=True)
        ty = np.cross(n, tx)
        d = tx * (rr * np.cos(ph))[:, None] + ty * (rr * np.sin(ph))[:, None] + n * np.sqrt(np.maximum(0, 1 - u1))[:, None]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.concatenate([p, d], 1).astype(F32)
        out.append(rays)
    allr = np.concatenate(out, 0)
    rng.shuffle(allr, axis=0)
    return np.ascontiguousarray(allr, F32)


def shadow_rays_toward(points, lo, hi, seed=11):
    """shadow rays from `points` toward uniformly sampled points in the box [lo, hi] (a light),
    light_dist = distance (trace.h:253-260,:337)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tgt = rng.random((len(points), 3)) * (np.asarray(hi, np.float64) - np.asarray(lo, np.float64)) + np.asarray(lo, np.float64)
    d = tgt - points.astype(np.float64)
    dist = np.linalg.norm(d, axis=1)
    dist[dist == 0] = 1
    d /= dist[:, None]
    return np.concatenate([points, d], 1).astype(F32), dist.astype(F32)


def procedural_texture(width, height, seed=1, is_float=False):
    """A seeded RGB texture: coloured checker with noise (uint8), or a sky with a bright lobe for an environment map
    (float32 texels in the 0..255-and-above range getTexColor divides by 255, as the reference's EXR maps are read)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    if not is_float:
        check = (((x * 8) // width + (y * 8) // height) % 2).astype(np.float32)
        base = np.stack([0.25 + 0.7 * check, 0.3 + 0.5 * (x / width), 0.9 - 0.6 * check], -1)
        img = np.clip(base * 255.0 + rng.integers(-20, 21, (height, width, 3)), 0, 255).astype(np.uint8)
        return np.ascontiguousarray(img)
    v = y / max(height - 1, 1)
    sky = np.stack([90 + 100 * (1 - v), 130 + 90 * (1 - v), 200 + 55 * (1 - v)], -1)
    sun = np.exp(-(((x - 0.3 * width) / (0.04 * width)) ** 2 + ((y - 0.25 * height) / (0.06 * height)) ** 2))
    img = sky + 6000.0 * sun[..., None] * np.array([1.0, 0.9, 0.7])
    return np.ascontiguousarray(img.astype(np.float32))


def textured_scene(n_lat=24, n_lon=48, max_depth=4, rng_seed=5, seed=13, tex_size=64, env_size=(128, 64), env_intensity=1.0,
                   samples_in_flight=0, rows=None):
    """Stand-in for BASELINE.json configs[3] (textured meshes + environment map): an open floor of TRIANGLE objects, one
    SMOOTH_TRIANGLE blob with a textured MATTE material and one FLAT_TRIANGLE blob with a textured PLASTIC material (texcoords
    running past 1 so the wrap is exercised), lit by a lat-long float environment map behind EnvLight.transform =
    mat3_rotate_y(-0.76) (buildscene.h:515) and a small sphere light.  Returns (SceneInput, RenderDesc, pos, look)."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0.08, math.pi - 0.08, n_lat + 1, dtype=np.float64)
    ph = np.linspace(0.0, 2.0 * math.pi, n_lon, endpoint=False, dtype=np.float64)
    T, P = np.meshgrid(th, ph, indexing="ij")
    rad = 1.0 + 0.2 * np.sin(2 * T) * np.cos(3 * P) + 0.03 * rng.standard_normal(T.shape)
    pos = np.stack([rad * np.sin(T) * np.cos(P), rad * np.cos(T), rad * np.sin(T) * np.sin(P)], -1).reshape(-1, 3).astype(F32)
    nrm = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    tan = np.stack([-np.sin(P), np.zeros_like(P), np.cos(P)], -1).reshape(-1, 3).astype(F32)
    uv = np.stack([2.5 * P / (2.0 * math.pi) - 0.25, 1.5 * T / math.pi], -1).reshape(-1, 2).astype(F32)     # beyond [0, 1] on purpose
    idx = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
            c, d = a + n_lon, b + n_lon
            idx += [a, c, b, b, c, d]
    idx = np.asarray(idx, np.int32)
    tri_uv = uv[idx].reshape(-1, 6)
    b1 = api.bake_mesh(pos, idx, scaling=(0.3, 0.35, 0.3), orientation=(10.0, 30.0, -5.0), location=(-0.45, 0.36, 0.0),
                       normals=np.asarray(nrm, F32), tangents=tan)
    b2 = api.bake_mesh(pos, idx, scaling=(0.28, 0.25, 0.3), orientation=(0.0, -40.0, 15.0), location=(0.45, 0.27, 0.1),
                       normals=np.asarray(nrm, F32), tangents=tan)
    s = 3.0
    floor = np.array([[[-s, 0, -s], [s, 0, -s], [s, 0, s]], [[-s, 0, -s], [s, 0, s], [-s, 0, s]]], F32).reshape(2, 9)
    nw, n1, n2 = len(floor), len(b1["tri_verts"]), len(b2["tri_verts"])
    tris = np.concatenate([floor, b1["tri_verts"], b2["tri_verts"]], 0)
    sl = ShapeList()
    sl.sphere((0.0, 1.6, 0.9), 0.08)
    shapes = np.array(sl.shapes, api.SHAPE_DTYPE)
    obj_type = np.concatenate([np.full(nw, api.TRIANGLE), np.full(n1, api.SMOOTH_TRIANGLE), np.full(n2, api.FLAT_TRIANGLE), [api.SPHERE]]).astype(np.int32)
    obj_ref = np.concatenate([np.arange(nw + n1 + n2), [0]]).astype(np.int32)
    si = api.SceneInput(tris, shapes, obj_type, obj_ref)
    pad3, pad9, pad6 = np.zeros((nw, 3), F32), np.zeros((nw, 9), F32), np.zeros((nw, 6), F32)
    ms = api.MeshShading(flat_normal=np.concatenate([pad3, b1["flat_normal"], b2["flat_normal"]], 0),
                         vertex_normals=np.concatenate([pad9, b1["vertex_normals"], b2["vertex_normals"]], 0),
                         vertex_tangents=np.concatenate([pad9, b1["vertex_tangents"], b2["vertex_tangents"]], 0),
                         normal_mat_index=np.concatenate([np.zeros(nw + n1, np.int32), np.ones(n2, np.int32)]),
                         normal_mats=np.stack([b1["normal_mat"].reshape(9), b2["normal_mat"].reshape(9)]),
                         vertex_uv=np.concatenate([pad6, tri_uv, tri_uv], 0))
    mats = np.array([api.material(api.MAT_MATTE, (0.55, 0.55, 0.5)), api.material(api.MAT_MATTE, (1.0, 1.0, 1.0)),
                     api.material(api.MAT_PLASTIC, (1.0, 1.0, 1.0), ks=(0.25, 0.25, 0.25), uroughness=0.15),
                     api.material(api.MAT_EMISSIVE, (1.0, 0.95, 0.9), 40.0)], api.MATERIAL_DTYPE)
    mat_of = [0] * nw + [1] * n1 + [2] * n2 + [3]
    textures = [procedural_texture(tex_size, tex_size, seed=seed), procedural_texture(tex_size // 2, tex_size, seed=seed + 1),
                procedural_texture(env_size[0], env_size[1], seed=seed + 2, is_float=True)]
    st, ct = np.float32(math.sin(np.float32(-0.76))), np.float32(math.cos(np.float32(-0.76)))      # mat3_rotate_y, util/mat.h:229-238
    xf = np.array([[ct, 0, st], [0, 1, 0], [-st, 0, ct]], F32)
    desc = api.RenderDesc(si, mat_of, mats, max_depth, seed=rng_seed, env=((1.0, 1.0, 1.0), env_intensity, 6.0), mesh_shading=ms,
                          textures=textures, material_diffuse_tex=[-1, 0, 1, -1], env_map=2, env_transform=xf,
                          samples_in_flight=samples_in_flight, rows=rows)
    return si, desc, (0.0, 0.9, 2.6), (0.0, 0.35, 0.0)
