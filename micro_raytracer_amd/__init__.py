"""micro_raytracer_amd — MI355X (gfx950) path-tracing backend behind micro-raytracer's Sampler API.

The product is csrc/ (hand-written HIP kernels + the C ABI of include/mrt.h, built into
libmrt_hip.so).  The Python modules are host-side plumbing: the reference's JSON scene schema
(scene.py), a mirror of its Sampler (sampler.py) and the multi-GPU row sharding (dist.py).
"""
from .scene import Render, load_render  # noqa: F401
from .sampler import Sampler, raytrace  # noqa: F401
from ._lib import MrtError  # noqa: F401
