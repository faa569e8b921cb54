"""SURVEY 8f N2, the geometry part: crt_obj_load against the reference's own loadOBJ (compiled into oracle/_ref, marked
`ref`) on synthetic OBJ files that exercise its record kinds and quirks; crt_mesh_normals_tangents (calcTriangleNormals'
vertex normals + calcTangentVec -- both live in buildscene.h, which cannot be compiled: parity unpinned) against a numpy
restatement; and the whole chain OBJ -> normals / tangents -> crt_bake_mesh -> scene input."""
import ctypes as C

import numpy as np
import pytest

from craytracer_amd import api
import oraclelib as ol

F32 = np.float32

OBJ_TEXT = """# a comment
mtllib nothing.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 0.5 1.25e0
   v -1.5 2 3    \r
vn 0 0 1
vn 0 1 0
vn 1 0 0
vt 0 0
vt 1 0
vt 1 1
vt 0.25 0.75
usemtl first
f 1 2 3 4
f 1/1/1 2/2/1 5/3/2
g second_group
f 2//3 3//3 5//2 
f -1 -2 -3
f 3/4 4/1 5/2
usemtl second
f 1 2 5
f 1 2 5
f 0 2 5 6
o object
f 4 5 6
f 6 5 4 3 2 1
"""


def write_obj(tmp_path, text=OBJ_TEXT, name="mesh.obj"):
    p = tmp_path / name
    p.write_bytes(text.encode())
    return str(p)


def test_obj_records_and_quirks(tmp_path):
    shapes = api.obj_load(write_obj(tmp_path))
    assert [s["mat_name"] for s in shapes] == ["first", "first", "", "second"]      # the shape closed by `o` gets no name
    s0 = shapes[0]                                  # closed by `g`: the quad (fan of two) and the v/vt/vn triangle
    assert s0["indices"].tolist() == [[0, 1, 2], [0, 2, 3], [4, 5, 6]]
    assert len(s0["positions"]) == 7                # (1,0,0) and (1,1,1) are different vertices: different triples
    assert len(s0["normals"]) == 3 and len(s0["texcoords"]) == 3
    total = sum(len(s["indices"]) for s in shapes)
    assert total == 2 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 4     # `f 0 2 5 6` loses its first vertex and stays a triangle
    # a last line without a newline is not read (OBJGetLine returns false at EOF before the loop body)
    a = api.obj_load(write_obj(tmp_path, "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nf 3 2 1", "cut.obj"))
    assert sum(len(s["indices"]) for s in a) == 1
    with pytest.raises(api.CrtError):
        api.obj_load(str(tmp_path / "missing.obj"))
    with pytest.raises(api.CrtError):               # a face that points past the vertex list
        api.obj_load(write_obj(tmp_path, "v 0 0 0\nv 1 0 0\nf 1 2 9\n", "bad.obj"))


@pytest.mark.ref
def test_obj_against_the_reference_loader(tmp_path):
    R = ol.ref()
    R.ref_obj_open.argtypes = [C.c_char_p]
    R.ref_obj_shape_sizes.argtypes = [C.c_int, C.c_void_p]
    R.ref_obj_shape_copy.argtypes = [C.c_int] + [C.c_void_p] * 5
    rng = np.random.default_rng(3)
    texts = [OBJ_TEXT]
    lines = []                                      # a bigger random one: shared and repeated vertices, all triple forms
    nv, nn, nt = 200, 60, 80
    lines += ["v %.6g %.6g %.6g" % tuple(rng.standard_normal(3)) for _ in range(nv)]
    lines += ["vn %.6g %.6g %.6g" % tuple(rng.standard_normal(3)) for _ in range(nn)]
    lines += ["vt %.6g %.6g" % tuple(rng.random(2)) for _ in range(nt)]
    for k in range(600):
        if k % 97 == 0:
            lines.append("usemtl mat%d" % k)
        if k % 211 == 0:
            lines.append("g group%d" % k)
        cnt = int(rng.integers(3, 6))
        form = int(rng.integers(0, 4))
        toks = []
        for _ in range(cnt):
            v, n, t = int(rng.integers(1, nv + 1)), int(rng.integers(1, nn + 1)), int(rng.integers(1, nt + 1))
            if rng.random() < 0.2:
                v = v - nv - 1                       # relative index
            toks.append([str(v), "%d/%d" % (v, t), "%d//%d" % (v, n), "%d/%d/%d" % (v, t, n)][form])
        lines.append("f " + " ".join(toks))
    texts.append("\n".join(lines) + "\n")
    import os
    # the reference copies the path up to its first '.' into a 64-byte field with strcpy (objloader.h:859): a long
    # temporary path overruns it and corrupts its locals, so the files are opened by a short relative name
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        _compare_with_reference(R, texts, tmp_path)
    finally:
        os.chdir(cwd)


def _compare_with_reference(R, texts, tmp_path):
    for k, text in enumerate(texts):
        write_obj(tmp_path, text, "m%d.obj" % k)
        path = "m%d.obj" % k
        mine = api.obj_load(path)
        n = R.ref_obj_open(path.encode())
        assert n == len(mine)
        for i, s in enumerate(mine):
            sizes = (C.c_int * 4)()
            R.ref_obj_shape_sizes(i, sizes)
            assert list(sizes) == [s["positions"].size, s["normals"].size, s["texcoords"].size, s["indices"].size]
            pos, nrm, tex = (np.zeros(max(sizes[j], 1), F32) for j in range(3))
            idx = np.zeros(max(sizes[3], 1), np.int32)
            mat = C.create_string_buffer(64)
            R.ref_obj_shape_copy(i, pos.ctypes.data, nrm.ctypes.data, tex.ctypes.data, idx.ctypes.data, mat)
            assert ol.same_bits(pos[:sizes[0]], s["positions"].ravel()) and ol.same_bits(nrm[:sizes[1]], s["normals"].ravel())
            assert ol.same_bits(tex[:sizes[2]], s["texcoords"].ravel()) and np.array_equal(idx[:sizes[3]], s["indices"].ravel())
            if not (k == 0 and i == 2):              # the shape closed by `o` carries an uninitialised name in the reference
                assert mat.value.decode() == s["mat_name"]


def np_normalize(v):
    length = F32(np.sqrt(F32(F32(F32(v[0] * v[0]) + F32(v[1] * v[1])) + F32(v[2] * v[2]))))
    return np.zeros(3, F32) if length == 0 else (v * F32(F32(1.0) / length)).astype(F32)


def test_generated_normals_and_tangents():
    rng = np.random.default_rng(5)
    pos = rng.standard_normal((40, 3)).astype(F32)
    idx = rng.integers(0, 40, (90, 3)).astype(np.int32)
    idx = idx[(idx[:, 0] != idx[:, 1]) & (idx[:, 1] != idx[:, 2]) & (idx[:, 0] != idx[:, 2])]
    nrm, tan = api.mesh_normals_tangents(pos, idx)
    want_n = np.zeros_like(pos)
    want_t = np.zeros_like(pos)
    for a, b, c in idx:
        e1, e2 = (pos[b] - pos[a]).astype(F32), (pos[c] - pos[a]).astype(F32)
        cr = np.array([F32(e1[1] * e2[2]) - F32(e1[2] * e2[1]), F32(e1[2] * e2[0]) - F32(e1[0] * e2[2]), F32(e1[0] * e2[1]) - F32(e1[1] * e2[0])], F32)
        fn = np_normalize(cr)
        for v in (a, b, c):
            want_n[v] = (want_n[v] + fn).astype(F32)
            want_t[v] = (want_t[v] + e1).astype(F32)       # u_axis = q0 * 1 - q1 * 0
    for v in range(len(pos)):
        want_n[v] = np_normalize(want_n[v])
        d = F32(F32(F32(want_n[v][0] * want_t[v][0]) + F32(want_n[v][1] * want_t[v][1])) + F32(want_n[v][2] * want_t[v][2]))
        want_t[v] = np_normalize((want_t[v] - (want_n[v] * d).astype(F32)).astype(F32))
    assert ol.same_bits(nrm, want_n) and ol.same_bits(tan, want_t)
    given = rng.standard_normal((40, 3)).astype(F32)
    n2, t2 = api.mesh_normals_tangents(pos, idx, given)
    assert ol.same_bits(n2, given) and np.isfinite(t2).all()
    with pytest.raises(api.CrtError):
        api.mesh_normals_tangents(pos, [[0, 1, 40]])


def test_obj_to_scene_input(tmp_path):
    """OBJ -> normals / tangents -> crt_bake_mesh -> a SceneInput of SMOOTH_TRIANGLE objects the builder accepts."""
    shapes = api.obj_load(write_obj(tmp_path))
    s = shapes[0]
    nrm, tan = api.mesh_normals_tangents(s["positions"], s["indices"])
    baked = api.bake_mesh(s["positions"], s["indices"], (2.0, 1.0, 0.5), (10.0, 20.0, 30.0), (1.0, 2.0, 3.0), normals=nrm, tangents=tan)
    T = len(s["indices"])
    assert baked["tri_verts"].shape == (T, 9) and baked["vertex_normals"].shape == (T, 9) and np.isfinite(baked["tri_verts"]).all()
    si = api.SceneInput(baked["tri_verts"], None, np.full(T, api.SMOOTH_TRIANGLE, np.int32), np.arange(T, dtype=np.int32))
    built = api.BuiltScene(si, api.ACCEL_BVH4)
    assert built.desc.num_obj == T
