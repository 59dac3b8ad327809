"""MI355X-native path-tracing core: host-side mirror of the reference's Scene / renderer facade over librt_hip.so.

The directory is named `ray-tracer_amd`; import it as `ray_tracer_amd` (the repository root carries a small alias module).
"""
from . import api, loaders, scene, scenes  # noqa: F401
from .api import load_library  # noqa: F401
from .renderer import HipRenderer  # noqa: F401
from .scene import Scene  # noqa: F401
