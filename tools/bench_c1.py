"""BASELINE.json configs[0] as a render (cornell_box.txt, 512 x 512, 16 spp, uniform grid), timed: tools/bench_c1.py [repeats]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from craytracer_amd import api, scenes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
accel = {"grid": api.ACCEL_GRID, "bvh4": api.ACCEL_BVH4}[sys.argv[2] if len(sys.argv) > 2 else "grid"]
si, desc = scenes.cornell29_render(max_depth=5, seed=16)
cam, flm = scenes.cornell_camera(512, 512)
scene = api.Scene(api.BuiltScene(si, accel))
r = api.Renderer(scene, desc, cam, flm)
r.render(16)
ts = []
for _ in range(reps):
    r.reset(0)
    t0 = time.perf_counter()
    r.render(16)
    ts.append(time.perf_counter() - t0)
st = r.stats()
print("C1 render 512x512x16: best %.2f ms, median %.2f ms (%.0f samples/s/pixel); extend %.2f ms shadow %.2f ms per render (kernel events)"
      % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3, 16 / sorted(ts)[len(ts) // 2], st["ms_extend"] / (reps + 1), st["ms_shadow"] / (reps + 1)))
r.close()
