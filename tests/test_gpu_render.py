"""crt_renderer_* (wavefront path tracing, SURVEY 8f N1 restricted form) on the GPU against
oracle_render, the CPU restatement of pathTrace / estimateDirect (PARITY UNPINNED for trace.h's
control flow, see oracle/crt_oracle_trace.cpp), with the same counter-based random numbers.

The bar is bit-identical images.  Both sides do the same float32 / float64 operations in the same
order (no contraction), sqrt and divide round correctly on both, and the one libm call on the path,
cosf / sinf in the sampling maps, is glibc's algorithm restated for the device (crt_libm.h, checked
against the platform libm on every float by test_libm.py).  Nothing weaker would hold: the reference
offsets no ray, so one different last bit in a hit point flips a self-shadowing decision and changes
the pixel visibly (with the device's own cosf / sinf, 27 % of the pixels of the first test differ).
"""
import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol

pytestmark = pytest.mark.gpu


def gpu_render(si, desc, cam, flm, accel, passes):
    built = api.BuiltScene(si, accel)
    scene = api.Scene(built)
    r = api.Renderer(scene, desc, cam, flm)
    for n in passes:
        r.render(n)
    buf, ns = r.read()
    st = r.stats()
    r.close()
    scene.close()
    built.close()
    return buf, ns, st


def compare(got, want):
    assert np.isfinite(got).all() and got.max() > 0
    assert ol.same_bits(got, want), "%.5f of the pixels are bit-identical" % (got == want).all(axis=2).mean()


@pytest.mark.parametrize("accel", ["NONE", "BVH4", "BVH", "GRID"])
def test_cornell_box_against_the_oracle(accel):
    """S1 with its materials (cornell_box.txt), 96 x 96, 8 samples, depth 5.  Through the accelerators the
    flat rectangles are partly missed, by the reference too (test_gpu_configs); the images still agree."""
    si, desc = scenes.cornell29_render(max_depth=5, seed=11)
    cam, flm = scenes.cornell_camera(96, 96)
    got, ns, st = gpu_render(si, desc, cam, flm, api.ACCEL_NAMES[accel], [8])
    assert ns == 8 and st["samples"] == 8 and st["rays_closest"] >= 8 * 96 * 96
    want = ol.OracleScene(si, api.ACCEL_NAMES[accel]).render(desc, cam, flm, 0, 8)
    compare(got, want)
    if accel == "NONE":
        light = want[..., 0] == np.float32(8 * 90.0)                 # pixels that see the light on every sample
        assert light.sum() > 20 and (got[light] == want[light]).all()


def test_room_with_every_light_kind():
    """Triangle mesh + matte spheres; rectangle, sphere and disk lights and a sampled constant
    environment light; QBVH."""
    si, desc, pos, look = scenes.render_room(48, 96, max_depth=6, rng_seed=5)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.2, 128, 96)
    got, ns, st = gpu_render(si, desc, cam, flm, api.ACCEL_BVH4, [6])
    want = ol.OracleScene(si, api.ACCEL_BVH4).render(desc, cam, flm, 0, 6)
    compare(got, want)
    assert st["rays_shadow"] > 0


def test_mirrors():
    """MIRROR: no direct light at the hit, specular reflection, and the emitted light / sky seen through
    it on the next bounce (sampled_flags & SPECULAR, trace.h:419)."""
    si, desc, pos, look = scenes.render_room(24, 48, max_depth=6, rng_seed=12, mirrors=True)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.2, 96, 96)
    got, _, _ = gpu_render(si, desc, cam, flm, api.ACCEL_BVH4, [4])
    want = ol.OracleScene(si, api.ACCEL_BVH4).render(desc, cam, flm, 0, 4)
    compare(got, want)
    _, plain, _, _ = scenes.render_room(24, 48, max_depth=6, rng_seed=12, mirrors=False)
    other = ol.OracleScene(si, api.ACCEL_BVH4).render(plain, cam, flm, 0, 4)
    assert not ol.same_bits(want, other)


def test_thin_transparent_surfaces():
    """TRANSPARENT (the reference's `thin` specular transmission): Fresnel pick between mirror reflection and
    straight-through transmission by one random number per bounce, no direct light at the hit."""
    si, desc, pos, look = scenes.render_room(24, 48, max_depth=6, rng_seed=21, mirrors=True, glass=True)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.2, 96, 96)
    got, _, _ = gpu_render(si, desc, cam, flm, api.ACCEL_BVH4, [4])
    want = ol.OracleScene(si, api.ACCEL_BVH4).render(desc, cam, flm, 0, 4)
    compare(got, want)


def test_oren_nayar_matte():
    """MATTE with sigma != 0: the Oren-Nayar lobe, evaluated in tangent space when sampled and -- as the
    reference does, reflection.cpp:731 -- on the world-space vectors in the direct-light estimate."""
    si, desc, pos, look = scenes.render_room(24, 48, max_depth=5, rng_seed=31, rough=True)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.2, 96, 96)
    got, _, _ = gpu_render(si, desc, cam, flm, api.ACCEL_BVH4, [4])
    osc = ol.OracleScene(si, api.ACCEL_BVH4)
    want = osc.render(desc, cam, flm, 0, 4)
    compare(got, want)
    _, smooth, _, _ = scenes.render_room(24, 48, max_depth=5, rng_seed=31, rough=False)
    assert not ol.same_bits(want, osc.render(smooth, cam, flm, 0, 4))


def test_plastic_glass_and_metal():
    """PLASTIC (two lobes: FresnelBlendDiffuse + FresnelBlendSpecular), GLASS (MicrofacetFresnel, rough, reflecting or
    refracting by one random number) and METAL (MicrofacetReflection with the conductor Fresnel term, one of them
    anisotropic), all Beckmann.  exp / log / tan / atan are double-precision results rounded once on both sides, so the
    images are expected to agree bit for bit here too; the bar written down is the weaker one the float functions
    justify: at most 1 pixel in 1 000 may differ."""
    si, desc, pos, look = scenes.render_room(24, 48, max_depth=6, rng_seed=41, glossy=True)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.2, 96, 96)
    got, _, _ = gpu_render(si, desc, cam, flm, api.ACCEL_BVH4, [4])
    osc = ol.OracleScene(si, api.ACCEL_BVH4)
    want = osc.render(desc, cam, flm, 0, 4)
    assert np.isfinite(got).all() and got.max() > 0
    same = (got == want).all(axis=2).mean()
    print("glossy room: %.5f of the pixels bit-identical" % same)
    assert same >= 0.999
    _, matte, _, _ = scenes.render_room(24, 48, max_depth=6, rng_seed=41)
    assert not ol.same_bits(want, osc.render(matte, cam, flm, 0, 4))


@pytest.mark.parametrize("smooth", [True, False])
def test_baked_mesh_with_interpolated_or_face_normals(smooth):
    """A mesh baked by crt_bake_mesh into SMOOTH_TRIANGLE objects (normals and tangents interpolated with the hit's
    barycentrics, through normal_mat) or FLAT_TRIANGLE objects (the transformed face normal, not turned towards the ray)."""
    si, desc, pos, look = scenes.smooth_blob_room(smooth=smooth)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.1, 96, 80)
    got, _, _ = gpu_render(si, desc, cam, flm, api.ACCEL_BVH4, [4])
    want = ol.OracleScene(si, api.ACCEL_BVH4).render(desc, cam, flm, 0, 4)
    compare(got, want)


def test_emissive_mesh_triangles_emit_but_are_not_sampled():
    """EMISSIVE on SMOOTH / FLAT triangles (the reference's MeshLights: no power, no case in estimateDirect): seen by camera
    rays, never chosen as a light."""
    for smooth in (True, False):
        si, desc, pos, look = scenes.smooth_blob_room(12, 24, smooth=smooth, max_depth=4)
        om = desc.obj_material.copy()
        first_blob = int(np.nonzero(si.obj_type != api.TRIANGLE)[0][0]) if (si.obj_type != api.TRIANGLE).any() else 12
        emissive = int(np.nonzero(desc.materials["type"] == api.MAT_EMISSIVE)[0][0])
        om[first_blob:first_blob + 200:3] = emissive             # every third triangle of the first 200 glows
        d2 = api.RenderDesc(si, om, desc.materials, 4, seed=9, env=None, mesh_shading=desc.mesh_shading)
        cam = api.make_camera(pos, look, focal_pt_dist=0.05)
        flm = api.make_film(cam, 1.2, 80, 64)
        got, _, _ = gpu_render(si, d2, cam, flm, api.ACCEL_BVH4, [3])
        want = ol.OracleScene(si, api.ACCEL_BVH4).render(d2, cam, flm, 0, 3)
        compare(got, want)


def test_scene_file_end_to_end():
    """tests/golden/scene_all.txt (this repository's own file in the reference's grammar: all seven material types, rectangle /
    sphere / triangle / disk objects, a smooth and a flat OBJ mesh, film, camera, env light) -> crt_scene_file_load -> builder
    -> preprocessLights' world radius -> renderer, against the oracle on the same description."""
    import os
    sf = api.SceneFile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_all.txt"))
    built = api.BuiltScene(sf.scene_input, api.ACCEL_BVH4)
    desc = sf.render_desc(5, built=built, seed=3)
    (pos, look), (_, _, w, h, fov) = sf.camera, sf.film
    cam = api.make_camera(pos, look)
    flm = api.make_film(cam, fov, w, h)
    scene = api.Scene(built)
    r = api.Renderer(scene, desc, cam, flm)
    r.render(4)
    got, _ = r.read()
    r.close(); scene.close()
    want = ol.OracleScene(sf.scene_input, built=built).render(desc, cam, flm, 0, 4)
    assert np.isfinite(got).all() and got.max() > 0
    same = (got == want).all(axis=2).mean()
    assert same >= 0.999, same


def test_flat_triangles_keep_their_winding_normal():
    si, desc, pos, look = scenes.render_room(16, 32, max_depth=3, rng_seed=9, tri_type=api.FLAT_TRIANGLE)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.2, 64, 64)
    got, _, _ = gpu_render(si, desc, cam, flm, api.ACCEL_BVH4, [4])
    want = ol.OracleScene(si, api.ACCEL_BVH4).render(desc, cam, flm, 0, 4)
    compare(got, want)


def test_passes_add_up_and_runs_repeat():
    """render(3) + render(5) == render(8) bit for bit (sample indices continue), also when the
    samples in flight differ; and a second run with the same seed repeats the first."""
    si, desc = scenes.cornell29_render(max_depth=4, seed=3)
    cam, flm = scenes.cornell_camera(64, 48)
    a, na, _ = gpu_render(si, desc, cam, flm, api.ACCEL_NONE, [8])
    b, nb, _ = gpu_render(si, desc, cam, flm, api.ACCEL_NONE, [3, 5])
    _, desc1 = scenes.cornell29_render(max_depth=4, seed=3, samples_in_flight=1)
    c, nc, _ = gpu_render(si, desc1, cam, flm, api.ACCEL_NONE, [8])
    assert na == nb == nc == 8
    assert ol.same_bits(a, b) and ol.same_bits(a, c)
    _, desc2 = scenes.cornell29_render(max_depth=4, seed=4)
    d, _, _ = gpu_render(si, desc2, cam, flm, api.ACCEL_NONE, [8])
    assert not ol.same_bits(a, d)


def test_sample_ranges_rendered_apart_add_up():
    """What two GPUs would do: samples [0, 4) and [4, 8) rendered separately (crt_renderer_reset) and
    summed equal one 8-sample render up to the order of the additions; each half is bit-identical
    to the oracle's same range."""
    si, desc = scenes.cornell29_render(max_depth=4, seed=6)
    cam, flm = scenes.cornell_camera(64, 48)
    built = api.BuiltScene(si, api.ACCEL_NONE)
    scene = api.Scene(built)
    r = api.Renderer(scene, desc, cam, flm)
    r.render(8)
    whole, _ = r.read()
    r.reset(0)
    r.render(4)
    lo, n_lo = r.read()
    r.reset(4)
    r.render(4)
    hi, n_hi = r.read()
    import torch
    d = torch.zeros(48 * 64 * 3, dtype=torch.float32, device="cuda")
    r.read_dev(d.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ol.same_bits(d.cpu().numpy().reshape(48, 64, 3), hi)
    r.close()
    assert n_lo == 4 and n_hi == 8
    osc = ol.OracleScene(si, api.ACCEL_NONE)
    assert ol.same_bits(lo, osc.render(desc, cam, flm, 0, 4))
    assert ol.same_bits(hi, osc.render(desc, cam, flm, 4, 4))
    assert np.allclose(lo + hi, whole, rtol=1e-5, atol=1e-6)


def test_the_renderer_leaves_the_scene_s_binning_setting_alone():
    """The renderer's own queries go out unbinned (their queues are in path order); that is decided per launch, not by
    switching the scene's setting: the setting reads the same before and after a render, on or off, and a caller's
    batch through the same scene is binned or not as the caller chose -- with identical hit records either way."""
    si, desc, pos, look = scenes.courtyard_render(4, 32, 64, max_depth=3, rng_seed=5)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.2, 512, 512)              # 2^18 pixels: the queries are large enough to be binned
    built = api.BuiltScene(si, api.ACCEL_BVH4)
    scene = api.Scene(built)
    rays = scenes.random_rays(1 << 18, *scenes.scene_bounds(si), seed=3)
    images, hits = [], []
    for on in (True, False):
        scene.set_ray_binning(on)
        assert scene.ray_binning() is on
        r = api.Renderer(scene, desc, cam, flm)
        r.render(2)
        assert scene.ray_binning() is on
        assert scene.last_binning_ms() == 0.0            # this thread's last launch was the renderer's: not binned
        images.append(r.read()[0])
        r.close()
        hits.append(scene.intersect(rays).copy())
        assert (scene.last_binning_ms() > 0.0) is on     # the caller's batch: as the caller chose
    scene.close()
    built.close()
    assert ol.same_bits(images[0], images[1])
    assert (hits[0].view(np.uint8) == hits[1].view(np.uint8)).all()


def test_depth_zero_is_emission_only():
    si, desc = scenes.cornell29_render(max_depth=0, seed=1)
    cam, flm = scenes.cornell_camera(64, 64)
    got, _, st = gpu_render(si, desc, cam, flm, api.ACCEL_NONE, [2])
    want = ol.OracleScene(si, api.ACCEL_NONE).render(desc, cam, flm, 0, 2)
    assert ol.same_bits(got, want)
    assert st["rays_shadow"] == 0 and st["rays_closest"] == 2 * 64 * 64
    vals = np.unique(got[..., 0])
    assert set(vals.tolist()) <= {0.0, 90.0, 180.0}


def test_unsupported_scenes_are_refused():
    si, desc = scenes.cornell29_render()
    cam, flm = scenes.cornell_camera(8, 8)
    built = api.BuiltScene(si, api.ACCEL_NONE)
    scene = api.Scene(built)
    mats = desc.materials.copy()
    mats[1]["type"] = 128                                            # not a MatType
    with pytest.raises(api.CrtError) as e:
        api.Renderer(scene, api.RenderDesc(si, desc.obj_material, mats, 3), cam, flm)
    assert e.value.code == -4
    # materials that could drive the path throughput negative (where the reference paints the path BLUE / RED, trace.h:452-453,
    # :478-479): a negative colour, PLASTIC ks above 1
    mats = desc.materials.copy()
    mats[0]["color"][1] = -0.1
    with pytest.raises(api.CrtError) as e:
        api.Renderer(scene, api.RenderDesc(si, desc.obj_material, mats, 3), cam, flm)
    assert e.value.code == -4
    mats = desc.materials.copy()
    mats[0] = api.material(api.MAT_PLASTIC, (0.5, 0.5, 0.5), ks=(0.2, 1.5, 0.2), uroughness=0.2)
    with pytest.raises(api.CrtError) as e:
        api.Renderer(scene, api.RenderDesc(si, desc.obj_material, mats, 3), cam, flm)
    assert e.value.code == -4
    om = desc.obj_material.copy()
    om[3] = 99
    with pytest.raises(api.CrtError) as e:
        api.Renderer(scene, api.RenderDesc(si, om, desc.materials, 3), cam, flm)
    assert e.value.code == -1
    with pytest.raises(api.CrtError) as e:                           # a scene built from other objects
        other = scenes.courtyard(1, 8, 8)
        api.Renderer(api.Scene(api.BuiltScene(other, api.ACCEL_BVH4)), desc, cam, flm)
    assert e.value.code == -1
    s3, d3, _, _ = scenes.smooth_blob_room(8, 16)
    b3 = api.BuiltScene(s3, api.ACCEL_BVH4)
    sc3 = api.Scene(b3)
    with pytest.raises(api.CrtError) as e:                           # SMOOTH_TRIANGLE objects without their shading arrays
        api.Renderer(sc3, api.RenderDesc(s3, d3.obj_material, d3.materials, 3), cam, flm)
    assert e.value.code == -4
    s2 = scenes.analytic_scene()                                     # a shape table that points outside itself
    b2 = api.BuiltScene(s2, api.ACCEL_BVH4)
    sc2 = api.Scene(b2)
    s2b = api.SceneInput(s2.tri_verts, s2.shapes.copy(), s2.obj_type, s2.obj_ref)
    inst = np.nonzero(s2b.shapes["type"] == api.INSTANCED)[0][0]
    s2b.shapes["inner"][inst] = 10 ** 6
    with pytest.raises(api.CrtError) as e:
        api.Renderer(sc2, api.RenderDesc(s2b, np.zeros(s2.num_obj, np.int32), desc.materials, 3), cam, flm)
    assert e.value.code == -1


def seam_tolerant_compare(got, want, max_fraction):
    """Analytic shapes go through device libm (atan2f / acosf in the phi / theta windows: class T4): a ray within an ulp of a
    sphere, cylinder or torus seam may be decided differently, and every path through such a hit differs from there on.
    Everything else is bit-identical, so the bar is a bound on the FRACTION of differing pixels, and it is reported."""
    assert np.isfinite(got).all() and got.max() > 0
    differ = (got != want).any(axis=2)
    assert differ.mean() <= max_fraction, "%d of %d pixels differ" % (int(differ.sum()), differ.size)
    return int(differ.sum())


@pytest.mark.parametrize("accel", ["NONE", "BVH", "BVH4", "GRID"])
@pytest.mark.parametrize("rotated", [False, True])
def test_analytic_shapes_render(accel, rotated):
    """Plane, disk, instanced box / torus / open cylinder / solid cylinder (a compound), sphere, rectangle, triangles: every shape kind
    shaded, through every accelerator (reference control flow: instance boxes are not bounding).  Shading records of the
    shapes whose intersectors leave dpdu unset follow the conventions of crt_fill_shade_rec, shared with the oracle."""
    si, desc, pos, look = scenes.analytic_render(max_depth=4, rng_seed=31, rotated=rotated)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 0.9, 160, 96)
    got, ns, st = gpu_render(si, desc, cam, flm, api.ACCEL_NAMES[accel], [4])
    want = ol.OracleScene(si, api.ACCEL_NAMES[accel]).render(desc, cam, flm, 0, 4)
    n = seam_tolerant_compare(got, want, 2e-3)
    print("analytic render %s rotated=%s: %d differing pixels" % (accel, rotated, n))
    assert st["rays_shadow"] > 0


def test_analytic_shapes_with_glossy_materials():
    si, desc, pos, look = scenes.analytic_render(max_depth=5, rng_seed=8, glossy=True)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 0.9, 128, 80)
    got, _, _ = gpu_render(si, desc, cam, flm, api.ACCEL_BVH4, [3])
    want = ol.OracleScene(si, api.ACCEL_BVH4).render(desc, cam, flm, 0, 3)
    seam_tolerant_compare(got, want, 2e-3)
