"""examples/c_host.c: a host program in C over include/crt_gpu.h alone -- scene file -> builder -> upload -> ray queries from
host arrays -> hit -> ShadeRec -> the sample loop -> image state and PPM (the reference's main.cpp:187-368 with the hot path on
the GPU), and with -DWITH_RCCL the C-side ncclAllReduce of the colour buffers.  Compiled on the CPU; run on the GPU box as a
child process, its files compared with the Python mirror of the same calls and with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from craytracer_amd import api
import oraclelib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c_host.c")
EXE = os.path.join(ROOT, "examples", "c_host")
SCENE = os.path.join(ROOT, "tests", "golden", "scene_all.txt")


def test_it_compiles_as_c99_against_the_header_alone(tmp_path):
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", SRC,
                    "-o", str(tmp_path / "c_host.o")], check=True)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/include/rccl/rccl.h"), reason="no RCCL headers")
def test_the_rccl_variant_compiles_and_links(tmp_path):
    """the all-reduce leg cannot run on a one-GPU box; that it is valid C against <rccl/rccl.h> and links is checked here"""
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-DWITH_RCCL", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                    "-I/opt/rocm/include", SRC, "-L" + os.path.join(ROOT, "craytracer_amd", "csrc"), "-lcrt_gpu", "-L/opt/rocm/lib",
                    "-lrccl", "-lamdhip64", "-lm", "-o", str(tmp_path / "c_host_rccl")], check=True)
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(tmp_path / "c_host_rccl")], capture_output=True, text=True, check=True).stdout
    assert "ncclAllReduce" in syms and "crt_renderer_read_dev" in syms


def test_without_a_gpu_it_says_so_and_fails():
    """no CPU fallback anywhere: on a machine without a gfx950 device the program stops at crt_device_count"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert os.path.exists(EXE), "examples/c_host is built by __graft_entry__.build()"
    r = subprocess.run([EXE, SCENE, "/tmp/never_written"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no gfx950 device" in r.stderr


@pytest.mark.gpu
def test_the_c_program_writes_what_the_python_mirror_and_the_oracle_compute(tmp_path):
    assert os.path.exists(EXE), "examples/c_host is built by __graft_entry__.build()"
    out = str(tmp_path / "c")
    spp = 4
    r = subprocess.run([EXE, SCENE, out, str(spp)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert "rays 65536" in r.stdout and "spp" in r.stdout
    rays = np.fromfile(out + ".rays", np.float32).reshape(-1, 6)
    hits = np.fromfile(out + ".hits", api.HIT_DTYPE)
    sh = np.fromfile(out + ".shadow", np.float32).reshape(-1, 2)
    assert len(rays) == len(hits) == len(sh) == 1 << 16
    assert np.allclose(np.linalg.norm(rays[:, 3:], axis=1), 1.0, atol=1e-5)
    # the same calls through the ctypes mirror, and the CPU oracle on the same tree
    sf = api.SceneFile(SCENE)
    built = api.BuiltScene(sf.scene_input, api.ACCEL_BVH4, sah=True)
    sc = api.Scene(built, 0)
    assert ol.same_bits(hits, sc.intersect(rays))
    o = ol.OracleScene(sf.scene_input, built=built)
    ref = o.intersect(rays, nthreads=8)
    same = (hits["obj"] == ref["obj"]) & (hits["part"] == ref["part"])
    assert same.mean() > 0.999                                  # class T4 (analytic-shape seams) aside
    tri = same & np.isin(sf.scene_input.obj_type[np.maximum(hits["obj"], 0)], (api.TRIANGLE, api.FLAT_TRIANGLE, api.SMOOTH_TRIANGLE)) & (hits["obj"] >= 0)
    assert tri.sum() > 1000 and ol.same_bits(hits[tri], ref[tri])
    ld = np.fromfile(out + ".ld", np.float32)
    t, occ = sc.shadow(rays, ld)
    assert ol.same_bits(sh[:, 0], t) and np.array_equal(sh[:, 1] != 0, occ.astype(bool))
    # the image: the same renderer calls from Python give the same sums, and the state file is the reference's format
    desc = sf.render_desc(5, built=built, seed=3)
    (pos, look), (_, _, w, h, fov) = sf.camera, sf.film
    cam = api.make_camera(pos, look)
    flm = api.make_film(cam, fov, w, h)
    rr = api.Renderer(sc, desc, cam, flm)
    rr.render(spp)
    want, _ = rr.read()
    rr.close()
    buf, n_s = api.image_state_load(out + ".is")
    assert n_s == spp and buf.shape == (h, w, 3)
    assert ol.same_bits(np.asarray(buf, np.float32).reshape(-1), np.asarray(want, np.float32).reshape(-1))
    with open(out + ".ppm", "rb") as f:
        assert f.read(2) == b"P6"
    sc.close()
    o.close()
