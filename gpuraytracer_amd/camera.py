"""Camera matrix recipe of the reference's render loop (RayTracer.cpp:25-26, :30-31, :128-129)."""
import math

import numpy as np

DEFAULT_TRANSLATION = (0.0, 1.5, -3.0)          # RayTracer.cpp:25
DEFAULT_ROTATION = (-math.pi / 4.0, 0.0)        # RayTracer.cpp:26  (rot.x, rot.y)
DEFAULT_CLOCK = 95834783                        # the constant left beside clock() at Scene.cpp:447


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float64)


def camera_matrix(translation=DEFAULT_TRANSLATION, rotation=DEFAULT_ROTATION):
    """CAMERA = translate(T) * rotate(-rot.y, Y) * rotate(-rot.x, X), built in double and narrowed to
    float32 as glm::fmat4(glm::dmat4) does.  Returns 16 floats, column-major (the layout
    glProgramUniformMatrix4fv receives at Scene.cpp:446)."""
    t = np.eye(4, dtype=np.float64)
    t[:3, 3] = translation
    m = t @ _rot_y(-rotation[1]) @ _rot_x(-rotation[0])
    return np.ascontiguousarray(m.T.astype(np.float32).reshape(16))
