"""Synthetic large scene for measurements: a height field of n x n cells (2 triangles each) displaced by a
hash, over a floor, lit by two emissive quads, with three materials.  Written as Wavefront OBJ/MTL text so
that it goes through the same ingest (`host.load_wavefront`, the reference's conventions of
Scene.cpp:576-702) as the bundled models.  n = 724 gives 1 048 352 + 6 triangles.
"""
import os

import numpy as np


def _pcg(k):
    k = np.asarray(k, dtype=np.uint64)
    state = (k * np.uint64(747796405) + np.uint64(2891336453)) & np.uint64(0xFFFFFFFF)
    shift = (state >> np.uint64(28)) + np.uint64(4)
    word = (((state >> shift) ^ state) * np.uint64(277803737)) & np.uint64(0xFFFFFFFF)
    return ((word >> np.uint64(22)) ^ word) & np.uint64(0xFFFFFFFF)


def heights(n, seed=1, amplitude=0.25):
    """(n+1) x (n+1) vertex heights: three octaves of hashed value noise, smooth enough to shade."""
    h = np.zeros((n + 1, n + 1), dtype=np.float64)
    for octave, cell in enumerate((64, 16, 4)):
        m = n // cell + 2
        idx = np.arange(m * m, dtype=np.uint64).reshape(m, m) + np.uint64(seed * 7919 + octave * 104729)
        lattice = _pcg(idx).astype(np.float64) / 4294967296.0
        x = np.arange(n + 1) / cell
        i0 = np.floor(x).astype(int)
        f = x - i0
        f = f * f * (3 - 2 * f)
        rows = lattice[i0][:, i0] * (1 - f)[None, :] + lattice[i0][:, i0 + 1] * f[None, :]
        rows1 = lattice[i0 + 1][:, i0] * (1 - f)[None, :] + lattice[i0 + 1][:, i0 + 1] * f[None, :]
        h += (rows * (1 - f)[:, None] + rows1 * f[:, None]) * (amplitude / (1 << octave))
    return h


def write_displaced_grid(base_path, n=724, seed=1, extent=4.0):
    """Writes <base_path>.obj and <base_path>.mtl.  Returns the triangle count."""
    h = heights(n, seed)
    xs = np.linspace(-extent, extent, n + 1)
    X, Z = np.meshgrid(xs, xs, indexing="xy")
    V = np.stack([X, h, Z], axis=-1).reshape(-1, 3)
    # vertex normals from central differences of the height field
    d = 2 * extent / n
    hx = np.gradient(h, d, axis=1)
    hz = np.gradient(h, d, axis=0)
    N = np.stack([-hx, np.ones_like(h), -hz], axis=-1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    N = N.reshape(-1, 3)
    T = np.stack([(X + extent) / (2 * extent) * 8.0, (Z + extent) / (2 * extent) * 8.0], axis=-1).reshape(-1, 2)
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    a = (jj * (n + 1) + ii).reshape(-1) + 1            # OBJ indices are 1-based
    b, c, e = a + 1, a + (n + 1), a + (n + 2)
    # counter-clockwise seen from above (+y): (a, c, b) and (b, c, e)
    faces = np.concatenate([np.stack([a, c, b], axis=1), np.stack([b, c, e], axis=1)])
    stripe = ((ii // max(1, n // 8) + jj // max(1, n // 8)) % 2).reshape(-1)
    stripe = np.concatenate([stripe, stripe])
    nv = len(V)
    with open(base_path + ".mtl", "w") as f:
        f.write("newmtl Ground\nNs 600\nKd 0.7 0.6 0.5\nKs 0.3 0.3 0.3\nKe 0 0 0\nNi 1.0\n\n")
        f.write("newmtl Polished\nNs 60\nKd 0.2 0.3 0.8\nKs 0.9 0.9 0.9\nKe 0 0 0\nNi 1.45\n\n")
        f.write("newmtl Lamp\nNs 900\nKd 1 1 1\nKs 0 0 0\nKe 12 11 9\nNi 1.0\n\n")
    with open(base_path + ".obj", "w") as f:
        f.write("mtllib %s.mtl\n" % os.path.basename(base_path))
        np.savetxt(f, V, fmt="v %.6f %.6f %.6f")
        # floor and two lamps: 12 more vertices
        y0 = -0.5
        extra = np.array([[-2 * extent, y0, -2 * extent], [2 * extent, y0, -2 * extent], [2 * extent, y0, 2 * extent],
                          [-2 * extent, y0, 2 * extent],
                          [-1.5, 3.0, -1.5], [-0.5, 3.0, -1.5], [-0.5, 3.0, -0.5], [-1.5, 3.0, -0.5],
                          [1.0, 2.5, 1.0], [2.0, 2.5, 1.0], [2.0, 2.5, 2.0], [1.0, 2.5, 2.0]])
        np.savetxt(f, extra, fmt="v %.6f %.6f %.6f")
        np.savetxt(f, N, fmt="vn %.6f %.6f %.6f")
        f.write("vn 0 1 0\nvn 0 -1 0\n")
        np.savetxt(f, T, fmt="vt %.6f %.6f")
        f.write("vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n")
        nn, nt = len(N), len(T)
        for name, sel in (("Ground", stripe == 0), ("Polished", stripe == 1)):
            f.write("o %s\nusemtl %s\n" % (name, name))
            fc = faces[sel]
            txt = np.stack([fc[:, 0], fc[:, 0], fc[:, 0], fc[:, 1], fc[:, 1], fc[:, 1], fc[:, 2], fc[:, 2], fc[:, 2]], axis=1)
            np.savetxt(f, txt, fmt="f %d/%d/%d %d/%d/%d %d/%d/%d")
        f.write("o Floor\nusemtl Ground\n")
        v = nv + 1
        up, down = nn + 1, nn + 2
        t = nt + 1
        f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (v, t, up, v + 2, t + 2, up, v + 1, t + 1, up))
        f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (v, t, up, v + 3, t + 3, up, v + 2, t + 2, up))
        f.write("o Lamps\nusemtl Lamp\n")
        for q in (v + 4, v + 8):      # facing down
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (q, t, down, q + 1, t + 1, down, q + 2, t + 2, down))
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (q, t, down, q + 2, t + 2, down, q + 3, t + 3, down))
    return len(faces) + 6


def texture_layers(size, seed=2):
    """Synthetic albedo / normal / ARM layers (RGBA float32, size x size) from the same hash."""
    idx = np.arange(size * size, dtype=np.uint64).reshape(size, size)
    def noise(k):
        cell = max(1, size // 64)
        coarse = _pcg((idx // np.uint64(size) // np.uint64(cell)) * np.uint64(size) + (idx % np.uint64(size)) // np.uint64(cell)
                      + np.uint64(seed * 15485863 + k * 32452843))
        return (coarse.astype(np.float64) / 4294967296.0).astype(np.float32)
    one = np.ones((size, size), dtype=np.float32)
    albedo = np.stack([0.3 + 0.6 * noise(0), 0.3 + 0.6 * noise(1), 0.3 + 0.6 * noise(2), one], axis=-1)
    nrm = np.stack([0.5 + 0.1 * (noise(3) - 0.5), 0.5 + 0.1 * (noise(4) - 0.5), 0.9 + 0.1 * noise(5), one], axis=-1)
    arm = np.stack([0.7 + 0.3 * noise(6), 0.2 + 0.7 * noise(7), (noise(8) > 0.8).astype(np.float32), one], axis=-1)
    return albedo.astype(np.float32), nrm.astype(np.float32), arm.astype(np.float32)
