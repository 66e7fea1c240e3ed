"""Writes tests/golden/scenes/<name>.rtscene from the reference's bundled OBJ/MTL models.

Run in the build container (needs /root/reference):  python tests/golden/make_scene_fixtures.py
Each file holds the ordered TriangleModel / TriangleShading / Material arrays that
Scene::finalizeObjects (Scene.cpp:382-395) would upload, as produced by the numpy restatement in
oracle/scene_np.py.  The GPU box has no /root/reference, so tests and bench.py load these files.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import scene_np  # noqa: E402

REF = "/root/reference/res/models"
OUT = os.path.join(HERE, "scenes")


def main():
    os.makedirs(OUT, exist_ok=True)
    index = {}
    for name in ("cubes", "box", "sphere", "_ObiWan"):
        models, shadings, lights, materials, names = scene_np.load_wavefront(os.path.join(REF, name))
        path = os.path.join(OUT, name + ".rtscene")
        scene_np.write_rtscene(path, models, shadings, lights, materials)
        index[name] = {
            "count": int(len(models)), "lights": int(lights), "materials": names,
            "models_sha256": hashlib.sha256(models.tobytes()).hexdigest(),
            "shadings_sha256": hashlib.sha256(shadings.tobytes()).hexdigest(),
        }
        print(name, index[name]["count"], lights, names)
    with open(os.path.join(OUT, "index.json"), "w") as fh:
        json.dump(index, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
