"""Identity of the kernel sources a profile was taken on: sha256 over the files of csrc/ and include/ (sorted by name).  The GPU box has no
.git, so a commit id cannot be read there; this id can be computed on both sides.  bench.py compares the id stored in a profile with the
id of the tree it runs from and says so in the JSON line (`from_profile`), instead of quoting counters of another build silently."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id(root: str = ROOT) -> str:
    h = hashlib.sha256()
    for sub in ("tensornetworkquantumsimulator.jl_amd/csrc", "include"):
        d = os.path.join(root, sub)
        for name in sorted(os.listdir(d)):
            p = os.path.join(d, name)
            if os.path.isfile(p) and name.rsplit(".", 1)[-1] in ("hip", "hpp", "cpp", "h", "sh"):
                h.update(name.encode()); h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build_id())
