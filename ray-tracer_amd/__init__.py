"""ctypes bindings of the two native libraries of this package.

  librtnative.so  the gfx950 HIP implementation behind the C-ABI of include/rt_native.h
  librthost.so    the C++ host side above that ABI: scene/BVH builder and frame driver (host/rth_api.h)

The directory name contains a hyphen, so import it with
``importlib.import_module("ray-tracer_amd")`` (tests/conftest.py does).  Nothing here computes anything: every
call goes straight into native code, and importing fails loudly when a library has not been built
(``make`` at the repository root, or ``__graft_entry__.build()``).
"""
from .bindings import (  # noqa: F401
    DTYPES,
    Driver,
    MainConfig,
    Mesh,
    NativeError,
    RisConfig,
    RisDriver,
    RtContext,
    Scene,
    default_config,
    host_lib,
    native_lib,
    ris_default_config,
    world_to_local,
)
from .scenes import build_scene_s0, build_scene_instanced, build_scene_textured, procedural_textures, ASSET_DIR  # noqa: F401
