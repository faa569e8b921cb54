"""Primary rays: the oracle's restatement of camera.cpp against the reference's own
camera.cpp (oracle/_ref, build container only), the product's host set-up against the oracle, and
both against the committed golden vectors.  Bit-exact: glibc on every side here."""
import ctypes as C
import os

import numpy as np
import pytest

from craytracer_amd import api
import oraclelib as ol
import camera_cases as cc

HERE = os.path.dirname(os.path.abspath(__file__))


def setup(which, name):
    pos, look, up, fpd, lens, fov, w, h = cc.CASES[name]
    cam = (ol.oracle_camera if which == "oracle" else ol.ref_camera)(pos, look, up, fpd, lens)
    return cam, ol.film(which, cam, fov, w, h), w * h


@pytest.mark.ref
@pytest.mark.parametrize("name", sorted(cc.CASES))
def test_oracle_equals_reference_camera(name):
    if not ol.have_ref():
        pytest.skip("oracle/_ref is only built where /root/reference exists")
    oc, of, n = setup("oracle", name)
    rc, rf, _ = setup("ref", name)
    assert ol.struct_bytes(oc) == ol.struct_bytes(rc)
    assert ol.struct_bytes(of) == ol.struct_bytes(rf)
    m = min(n, 200000)
    first = n - m
    isamp, lsamp = cc.samples(name, m)
    assert ol.same_bits(ol.primary_rays("oracle", oc, of, first, m, isamp, lsamp), ol.primary_rays("ref", rc, rf, first, m, isamp, lsamp))
    assert ol.same_bits(ol.primary_rays("oracle", oc, of, 0, 4096), ol.primary_rays("ref", rc, rf, 0, 4096))


@pytest.mark.parametrize("name", sorted(cc.CASES))
def test_product_host_setup_equals_oracle(name):
    pos, look, up, fpd, lens, fov, w, h = cc.CASES[name]
    cam = api.make_camera(pos, look, up, fpd, lens)
    flm = api.make_film(cam, fov, w, h)
    oc, of, _ = setup("oracle", name)
    assert ol.struct_bytes(cam) == ol.struct_bytes(oc)
    assert ol.struct_bytes(flm) == ol.struct_bytes(of)


@pytest.mark.parametrize("name", sorted(cc.CASES))
def test_oracle_equals_golden_camera_vectors(name):
    g = np.load(os.path.join(HERE, "golden", "camera_" + name + ".npz"), allow_pickle=False)
    oc, of, n = setup("oracle", name)
    assert ol.struct_bytes(oc) == g["camera"].tobytes() and ol.struct_bytes(of) == g["film"].tobytes()
    m = len(g["rays"])
    isamp, lsamp = cc.samples(name, m)
    assert ol.same_bits(isamp, g["image_samples"]) and ol.same_bits(lsamp, g["lens_samples"])
    assert ol.same_bits(ol.primary_rays("oracle", oc, of, int(g["first_pixel"][0]), m, isamp, lsamp), g["rays"])
    assert ol.same_bits(ol.primary_rays("oracle", oc, of, 0, len(g["rays_center"])), g["rays_center"])


def test_bad_arguments():
    L = api.lib()
    cam = api.CrtCamera()
    assert L.crt_camera_init(None, 0, 0.0, 0.0) == -1
    assert L.crt_camera_init(C.byref(cam), 2, 0.0, 0.0) == -1
    flm = api.CrtFilm()
    assert L.crt_film_dimension(C.byref(flm), 1.0, 0, 4, C.byref(cam)) == -1
