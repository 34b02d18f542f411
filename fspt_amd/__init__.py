"""fspt_amd — MI355X-native path-trace hot path of apbodnar/FSPT.

Product code only: the HIP kernels + C ABI (csrc/, libfspt.so), the ctypes
binding (_lib), the host mirror of the reference's frame driver (tracer) and
the scene assembly helpers (scene).  Nothing here imports oracle/.
"""
from ._lib import FsptError, lib  # noqa: F401
from .scene import SceneArrays, build_scene, bunny_scene, BUNNY_CAMERA  # noqa: F401
from .tracer import MultiPathTracer, PathTracer, Scene, bytes_per_sample, device_memory, set_texture_interleave_budget  # noqa: F401
