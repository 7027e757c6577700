"""MI355X-native visibility-buffer hot path (cull -> raster -> deferred PBR shade) behind the C ABI
of ``include/rdr_abi.h``; see ``renderer.Renderer`` for the host-side mirror of the reference API."""
from .scene import Camera, RendererConfig, Scene  # noqa: F401
from .renderer import Renderer, RendererError, build_ubo, load_library  # noqa: F401
