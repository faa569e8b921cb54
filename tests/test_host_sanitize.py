"""The host code (crt_build.cpp, crt_obj.cpp, crt_mesh.cpp, crt_image.cpp, crt_scenefile.cpp) under AddressSanitizer + UndefinedBehaviorSanitizer: built with
g++ on the spot (host code only, no GPU) and run on awkward inputs and error paths."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_builder_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-ffp-contract=off", os.path.join(ROOT, "tests", "host_sanitize.cpp"),
                    os.path.join(ROOT, "craytracer_amd", "csrc", "crt_build.cpp"), os.path.join(ROOT, "craytracer_amd", "csrc", "crt_obj.cpp"),
                    os.path.join(ROOT, "craytracer_amd", "csrc", "crt_mesh.cpp"), os.path.join(ROOT, "craytracer_amd", "csrc", "crt_image.cpp"),
                    os.path.join(ROOT, "craytracer_amd", "csrc", "crt_scenefile.cpp"),
                    "-o", exe, "-lpthread"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "host_sanitize ok" in out.stdout
