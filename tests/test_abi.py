"""CPU checks of the drop-in boundary: the shared library loads without a GPU and exports exactly
the symbols include/gpurt.h declares.  No compute call is made here."""
import ctypes
import os
import re

import numpy as np

from conftest import ROOT
import gpuraytracer_amd as g
from gpuraytracer_amd import api


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpurt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(api.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(g.lib_path())
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_record_layouts_match_header():
    assert g.TRI_MODEL_DTYPE.itemsize == 48
    assert g.TRI_SHADING_DTYPE.itemsize == 100
    assert g.MATERIAL_DTYPE.itemsize == 48
    assert g.TRI_SHADING_DTYPE.fields["normals"][1] == 4 and g.TRI_SHADING_DTYPE.fields["tex_p"][1] == 76
    assert ctypes.sizeof(api.RtStats) == 184


def test_null_context_is_rejected_without_a_gpu():
    lib = api.load_library()
    assert lib.rt_trace(None, 0, 1, 1) == -1
    assert lib.rt_resize(None, 8, 8) == -1
    assert lib.rt_last_error(None) == b"null context"
    h = ctypes.c_void_p()
    assert lib.rt_create(-1, ctypes.byref(h)) < 0 and not h.value


def test_camera_matrix_recipe():
    # RayTracer.cpp:25-26,30-31: T = (0, 1.5, -3), rot = (-pi/4, 0): camera looks 45 degrees down, +z forward
    m = g.camera_matrix().reshape(4, 4).T
    s = np.float32(np.sqrt(0.5))
    assert np.allclose(m[:3, 3], [0, 1.5, -3])
    assert np.allclose(m[:3, 2], [0, -s, s], atol=1e-7)
    assert np.allclose(m[:3, :3] @ m[:3, :3].T, np.eye(3), atol=1e-6)


def test_rtscene_roundtrip(tmp_path):
    from conftest import load_scene
    models, shadings, lights, materials = load_scene("cubes")
    p = str(tmp_path / "x.rtscene")
    g.write_rtscene(p, models, shadings, lights, materials)
    m2, s2, l2, mat2 = g.read_rtscene(p)
    assert l2 == lights and m2.tobytes() == models.tobytes() and s2.tobytes() == shadings.tobytes()
    assert mat2.tobytes() == materials.tobytes()
