"""craytracer_amd -- MI355X (gfx950) ray traversal + intersection behind craytracer's
scene-intersection API.  The product is csrc/libcrt_gpu.so (hand-written HIP kernels + a C ABI,
include/crt_gpu.h); this package is the thin host-side binding used by tests and bench.py."""
from . import api, scenes  # noqa: F401
from .api import (ACCEL_BVH, ACCEL_BVH4, ACCEL_GRID, ACCEL_NONE, BuiltScene, Scene, SceneInput,  # noqa: F401
                  TRAVERSE_AUTO, TRAVERSE_EXACT, TRAVERSE_PRUNED)
