"""MI355X-native path-tracing core: ctypes access to libgpurt.so (the C ABI in include/gpurt.h).

The compute path is the HIP library only.  Importing this package does not load it; creating a
`Context` does, and raises if the library has not been built (python -c "import __graft_entry__ as g; g.build()").
"""
from .api import Context, GpuRtError, TRI_MODEL_DTYPE, TRI_SHADING_DTYPE, MATERIAL_DTYPE, lib_path  # noqa: F401
from .camera import camera_matrix, DEFAULT_CLOCK  # noqa: F401
from .scene_io import read_rtscene, write_rtscene  # noqa: F401
