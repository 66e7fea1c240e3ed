"""CPU restatement of the reference's scene ingest and triangle pre-processing.

TEST INFRASTRUCTURE ONLY: used by tests/ and by the script that writes the scene
fixtures under tests/golden/scenes/.  Nothing in the product imports it.

Restates, in float32 numpy arithmetic (separate multiply and add, no fused operations):
  * Scene::readWFMaterial      Scene.cpp:659-702  (MTL: Kd, Ks, Ke, Ni, Ns/1000)
  * Scene::addWavefrontModel   Scene.cpp:585-657  (OBJ: one material per `o` block, last usemtl wins)
  * Triangle::Triangle         3Dobjects.hpp:39-55 (u, v, cross(u, v), per-vertex tangents)
  * Scene::createTrianglesBuffers Scene.cpp:283-340 (emissive triangles first in file order, the rest
    by descending |cross(u, v)|)

PARITY UNPINNED: the reference has no fixtures for these buffers.  One deliberate
difference: the reference sorts with std::sort (unstable, Scene.cpp:334), so the order of
equal-area triangles there depends on the C++ library; here ties keep file order.
"""
import ctypes
import struct

import numpy as np

f32 = np.float32

_libc = ctypes.CDLL(None)
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]

TRI_MODEL_DTYPE = np.dtype([("true_normal", "<f4", 3), ("position", "<f4", 3), ("u", "<f4", 3), ("v", "<f4", 3)])
TRI_SHADING_DTYPE = np.dtype(
    [("material_id", "<u4"), ("normals", "<f4", (3, 3)), ("tangents", "<f4", (3, 3)),
     ("tex_p", "<f4", 2), ("tex_u", "<f4", 2), ("tex_v", "<f4", 2)])
MATERIAL_DTYPE = np.dtype([("albedo", "<f4", 4), ("specular_roughness", "<f4", 4), ("emission_ior", "<f4", 4)])
assert TRI_MODEL_DTYPE.itemsize == 48 and TRI_SHADING_DTYPE.itemsize == 100 and MATERIAL_DTYPE.itemsize == 48


def _scan_floats(line, n):
    """sscanf(line, "%*s %f ..."): skip the first token, parse up to n floats with strtof."""
    toks = line.split()[1:1 + n]
    out = []
    for t in toks:
        try:
            float(t)
        except ValueError:
            break
        out.append(_libc.strtof(t.encode(), None))
    return out


def read_mtl(path):
    """Scene::readWFMaterial.  Returns (names, materials[MATERIAL_DTYPE])."""
    names, mats = [], []
    cur = None
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.rstrip("\n")
            if len(line) <= 3:
                continue
            if line[:6] == "newmtl":
                name = line[7:]
                if name in names:
                    cur = mats[names.index(name)]
                else:
                    names.append(name)
                    cur = np.zeros((), dtype=MATERIAL_DTYPE)
                    mats.append(cur)
            if cur is None:
                continue
            key = line[:2]
            if key == "Kd":
                v = _scan_floats(line, 3)
                cur["albedo"][:len(v)] = v
            elif key == "Ks":
                v = _scan_floats(line, 3)
                cur["specular_roughness"][:len(v)] = v
            elif key == "Ke":
                v = _scan_floats(line, 3)
                cur["emission_ior"][:len(v)] = v
            elif key == "Ni":
                v = _scan_floats(line, 1)
                if v:
                    cur["emission_ior"][3] = v[0]
            elif key == "Ns":
                v = _scan_floats(line, 1)
                if v:
                    cur["specular_roughness"][3] = f32(v[0]) / f32(1000.0)
    out = np.zeros(len(mats), dtype=MATERIAL_DTYPE)
    for i, m in enumerate(mats):
        out[i] = m
    return names, out


def read_obj(path, material_names):
    """Scene::addWavefrontModel.  Returns a list of objects, each (name, material_index, faces) where
    faces is an int array [n, 3, 3] of (pos, tex, nrm) 1-based indices, plus the global attribute arrays."""
    positions, normals, uvs = [], [], []
    objects = []
    cur = None
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.rstrip("\n")
            if len(line) <= 2:
                continue
            c0 = line[0]
            if c0 == "o":
                cur = {"name": line[2:], "material_index": None, "faces": []}
                objects.append(cur)
            elif c0 == "v":
                c1 = line[1]
                if c1 == " ":
                    v = _scan_floats(line, 3)
                    positions.append((v + [0.0, 0.0, 0.0])[:3])
                elif c1 == "t":
                    v = _scan_floats(line, 2)
                    uvs.append((v + [0.0, 0.0])[:2])
                elif c1 == "n":
                    v = _scan_floats(line, 3)
                    normals.append((v + [0.0, 0.0, 0.0])[:3])
            elif cur is not None and line[:6] == "usemtl":
                name = line[7:]
                cur["material_index"] = material_names.index(name) if name in material_names else -1
            elif c0 == "f":
                if cur is None:
                    continue          # the reference collects such faces and drops them at the next `o`
                toks = line.split()[1:4]
                cur["faces"].append([[int(x, 0) for x in t.split("/")] for t in toks])
    return (objects, np.array(positions, dtype=f32).reshape(-1, 3), np.array(uvs, dtype=f32).reshape(-1, 2),
            np.array(normals, dtype=f32).reshape(-1, 3))


def _cross(a, b):
    # glm::cross / GLSL cross: (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
    return np.stack([a[:, 1] * b[:, 2] - b[:, 1] * a[:, 2],
                     a[:, 2] * b[:, 0] - b[:, 2] * a[:, 0],
                     a[:, 0] * b[:, 1] - b[:, 0] * a[:, 1]], axis=1).astype(f32)


def _dot(a, b):
    # glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z
    t = (a * b).astype(f32)
    return ((t[:, 0] + t[:, 1]).astype(f32) + t[:, 2]).astype(f32)


def build_triangles(p0, p1, p2, n0, n1, n2, t0, t1, t2):
    """Triangle::Triangle for arrays of triangles.  Returns dict of float32 arrays."""
    u = (p1 - p0).astype(f32)
    v = (p2 - p0).astype(f32)
    true_normal = _cross(u, v)
    tex_u = (t1 - t0).astype(f32)
    tex_v = (t2 - t0).astype(f32)
    T = (tex_v[:, 0:1] * u - tex_u[:, 0:1] * v).astype(f32)
    tangents = []
    with np.errstate(all="ignore"):
        for N in (n0, n1, n2):
            w = (T - _dot(N, T)[:, None] * N).astype(f32)
            inv = (f32(1.0) / np.sqrt(_dot(w, w)).astype(f32)).astype(f32)   # glm::inversesqrt
            tangents.append((w * inv[:, None]).astype(f32))
    return {"position": p0.astype(f32), "u": u, "v": v, "true_normal": true_normal,
            "normals": np.stack([n0, n1, n2], axis=1).astype(f32),
            "tangents": np.stack(tangents, axis=1).astype(f32),
            "tex_p": t0.astype(f32), "tex_u": tex_u, "tex_v": tex_v}


def order_and_split(tri, material_ids, materials):
    """Scene::createTrianglesBuffers.  Returns (models, shadings, lights)."""
    n = len(material_ids)
    ke = materials["emission_ior"][:, :3]
    emissive_mat = _dot(ke, ke) > 0
    is_light = emissive_mat[material_ids]
    light_idx = np.nonzero(is_light)[0]
    rest_idx = np.nonzero(~is_light)[0]
    c = _cross(tri["u"][rest_idx], tri["v"][rest_idx])
    area_key = np.sqrt(_dot(c, c)).astype(f32)
    order_rest = rest_idx[np.argsort(-area_key.astype(np.float64), kind="stable")]
    order = np.concatenate([light_idx, order_rest]).astype(np.int64)
    models = np.zeros(n, dtype=TRI_MODEL_DTYPE)
    shadings = np.zeros(n, dtype=TRI_SHADING_DTYPE)
    models["true_normal"] = tri["true_normal"][order]
    models["position"] = tri["position"][order]
    models["u"] = tri["u"][order]
    models["v"] = tri["v"][order]
    shadings["material_id"] = material_ids[order]
    shadings["normals"] = tri["normals"][order]
    shadings["tangents"] = tri["tangents"][order]
    shadings["tex_p"] = tri["tex_p"][order]
    shadings["tex_u"] = tri["tex_u"][order]
    shadings["tex_v"] = tri["tex_v"][order]
    return models, shadings, int(len(light_idx))


def load_wavefront(base_path):
    """addWavefrontModel(base) + finalizeObjects: returns (models, shadings, lights, materials, names)."""
    names, materials = read_mtl(base_path + ".mtl")
    objects, pos, uv, nrm = read_obj(base_path + ".obj", names)
    faces, mat_ids = [], []
    for ob in objects:
        if not ob["faces"]:
            continue
        if ob["material_index"] is None or ob["material_index"] < 0:
            raise ValueError("object %r has no known material" % ob["name"])
        f = np.array(ob["faces"], dtype=np.int64)
        faces.append(f)
        mat_ids.append(np.full(len(f), ob["material_index"], dtype=np.uint32))
    faces = np.concatenate(faces)
    mat_ids = np.concatenate(mat_ids)
    pi, ti, ni = faces[:, :, 0] - 1, faces[:, :, 1] - 1, faces[:, :, 2] - 1
    tri = build_triangles(pos[pi[:, 0]], pos[pi[:, 1]], pos[pi[:, 2]],
                          nrm[ni[:, 0]], nrm[ni[:, 1]], nrm[ni[:, 2]],
                          uv[ti[:, 0]], uv[ti[:, 1]], uv[ti[:, 2]])
    models, shadings, lights = order_and_split(tri, mat_ids, materials)
    return models, shadings, lights, materials, names


# ------------------------------------------------------------------ scene fixture files

MAGIC = b"RTSC0001"


def write_rtscene(path, models, shadings, lights, materials):
    """Little-endian: magic, count, lights, n_materials, reserved, then the three record arrays."""
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<IIII", len(models), lights, len(materials), 0))
        fh.write(np.ascontiguousarray(models).tobytes())
        fh.write(np.ascontiguousarray(shadings).tobytes())
        fh.write(np.ascontiguousarray(materials).tobytes())


def read_rtscene(path):
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:8] != MAGIC:
        raise ValueError("not an rtscene file: %s" % path)
    count, lights, n_mat, _ = struct.unpack_from("<IIII", data, 8)
    off = 24
    models = np.frombuffer(data, dtype=TRI_MODEL_DTYPE, count=count, offset=off).copy()
    off += 48 * count
    shadings = np.frombuffer(data, dtype=TRI_SHADING_DTYPE, count=count, offset=off).copy()
    off += 100 * count
    materials = np.frombuffer(data, dtype=MATERIAL_DTYPE, count=n_mat, offset=off).copy()
    return models, shadings, lights, materials
