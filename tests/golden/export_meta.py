#!/usr/bin/env python
"""JSON sidecars of the golden fixtures for readers that cannot parse the `meta` string stored inside the .npz files (NPZ.jl reads
numeric arrays only): tests/golden/meta/<name>.json = the fixture's meta dictionary + `neighbors` (per vertex, the 0-based positions of
its neighbours in the leg order of the stored site tensors: ascending vertex position) + `arrays` (name -> shape).  julia/replay_golden.jl
reads these.  Run after make_golden.py:  python tests/golden/export_meta.py"""
import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    os.makedirs(os.path.join(HERE, "meta"), exist_ok=True)
    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        name = os.path.basename(path)[:-4]
        z = np.load(path)
        meta = json.loads(str(z["meta"])) if "meta" in z.files else {"name": name}
        if "vertices" in meta:
            pos = {tuple(v): i for i, v in enumerate(meta["vertices"])}
            nbrs = [[] for _ in meta["vertices"]]
            for a, b in meta["edges"]:
                nbrs[pos[tuple(a)]].append(pos[tuple(b)]); nbrs[pos[tuple(b)]].append(pos[tuple(a)])
            meta["neighbors"] = [sorted(n) for n in nbrs]
        meta["arrays"] = {k: list(z[k].shape) for k in z.files if k != "meta"}
        meta["array_dtypes"] = {k: str(z[k].dtype) for k in z.files if k != "meta"}
        with open(os.path.join(HERE, "meta", name + ".json"), "w") as f:
            json.dump(meta, f, indent=1)
        print("wrote meta/" + name + ".json")


if __name__ == "__main__":
    main()
