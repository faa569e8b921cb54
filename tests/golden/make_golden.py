#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref/libcrt_ref.so).

Runs only in the build container (needs /root/reference to build oracle/_ref).  The vectors are
data: seeded rays in, {t, obj, part, beta, gamma} and shadow {t, occluded} out, plus what the
reference's builders produced (final object order, QBVH / BVH nodes, grid cells).  Scenes are
regenerated from craytracer_amd.scenes by name; a checksum of the scene arrays is stored so that a
drifting generator is caught.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from craytracer_amd import api, scenes  # noqa: E402
import oraclelib as ol  # noqa: E402

SCENES = {
    "cornell29": (scenes.cornell29, ([-50, -50, -600], [600, 600, 50])),
    "analytic": (scenes.analytic_scene, ([-7, -2, -7], [7, 5, 7])),
    "analytic_unrotated": (lambda: scenes.analytic_scene(False), ([-7, -2, -7], [7, 5, 7])),
    "rand4k": (lambda: scenes.random_tris(4096, seed=17), ([-0.2] * 3, [1.2] * 3)),
}
N_RAYS = 4096


def scene_digest(si):
    h = hashlib.sha256()
    h.update(si.tri_verts.tobytes())
    h.update(si.shapes.tobytes())
    if si.obj_type is not None:
        h.update(si.obj_type.tobytes())
        h.update(si.obj_ref.tobytes())
    return h.hexdigest()


def golden_rays(name):
    lo, hi = SCENES[name][1]
    seed = int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")
    rays = scenes.random_rays(N_RAYS, np.array(lo, np.float32), np.array(hi, np.float32), seed)
    ax = scenes.axis_parallel_rays(256, np.array(lo, np.float32), np.array(hi, np.float32), seed + 1)
    rays = np.concatenate([rays[:-256], ax], 0)
    ld = (np.random.default_rng(seed + 2).random(N_RAYS) * np.linalg.norm(np.array(hi) - np.array(lo))).astype(np.float32)
    return np.ascontiguousarray(rays), ld


def camera_golden():
    """Primary rays by the reference's own camera.cpp: the last 4096 pixels with seeded samples and
    the first 4096 with the (0.5, 0.5) default."""
    import camera_cases as cc
    for name, (pos, look, up, fpd, lens, fov, w, h) in cc.CASES.items():
        cam = ol.ref_camera(pos, look, up, fpd, lens)
        flm = ol.film("ref", cam, fov, w, h)
        m = 4096
        first = w * h - m
        isamp, lsamp = cc.samples(name, m)
        np.savez_compressed(os.path.join(HERE, "camera_" + name + ".npz"),
                            camera=np.frombuffer(ol.struct_bytes(cam), np.uint8), film=np.frombuffer(ol.struct_bytes(flm), np.uint8),
                            first_pixel=np.array([first], np.int64), image_samples=isamp, lens_samples=lsamp,
                            rays=ol.primary_rays("ref", cam, flm, first, m, isamp, lsamp),
                            rays_center=ol.primary_rays("ref", cam, flm, 0, m))
        print("camera_" + name, "ok")


def main():
    assert ol.have_ref(), "build oracle/_ref first: make -C oracle ref"
    camera_golden()
    for name, (make, _) in SCENES.items():
        si = make()
        rays, ld = golden_rays(name)
        out = {"digest": np.frombuffer(scene_digest(si).encode(), np.uint8), "rays": rays, "light_dist": ld}
        for accel, tag in ((api.ACCEL_NONE, "none"), (api.ACCEL_BVH4, "bvh4")):
            ol.libc_srand(1)
            r = ol.RefScene(si, accel)
            hits = r.intersect(rays)
            t, occ = r.shadow(rays, ld)
            out["hits_" + tag] = hits
            out["shadow_t_" + tag] = t
            out["shadow_occ_" + tag] = occ
        for accel, tag in ((api.ACCEL_GRID, "grid"), (api.ACCEL_BVH, "bvh"), (api.ACCEL_BVH4, "bvh4")):
            ol.libc_srand(1)
            r = ol.RefScene(si, accel)
            ids, nng = r.order()
            out["order_" + tag] = ids
            out["nng_" + tag] = np.array([nng], np.int32)
            if accel == api.ACCEL_BVH4:
                out["nodes4"] = r.nodes4()
            elif accel == api.ACCEL_BVH:
                out["nodes2"] = r.nodes2()
            else:
                g, cs, ci = r.grid()
                out["grid_hdr"] = np.frombuffer(bytes(g), np.uint8).copy()
                out["cell_start"] = cs
                out["cell_items"] = ci
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), "bytes;", int((out["hits_none"]["obj"] >= 0).sum()), "hits (NONE),",
              int((out["hits_bvh4"]["obj"] >= 0).sum()), "hits (BVH4)")


if __name__ == "__main__":
    main()
