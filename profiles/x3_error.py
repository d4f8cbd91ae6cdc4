"""Error of the chi = 32 plane kernels against an f64 reference (numpy), for the route the environment selects:
    python profiles/x3_error.py            bf16 x 3 split on the bf16 matrix cores (kernels_x3.hip, default)
    TNQS_NO_BF16X3=1 python profiles/x3_error.py      f32 matrix instructions, three-multiplication product
    TNQS_NO_BF16X3=1 TNQS_NO_3M=1 ...                 f32 matrix instructions, four-multiplication product
prints max and rms error relative to the largest reference entry (what the tests bound) and relative to the rms entry."""
import ctypes as C
import json
import os
import sys
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "tensornetworkquantumsimulator.jl_amd", "libtnqs_hip.so"))
rng = np.random.default_rng(7)
chi = (32, 32, 32, 32)
z = 4
cchi = (C.c_int * z)(*chi)


def rnd(n, scale=1.0):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * scale).astype(np.complex64)


def site(scale=1.0):
    shp = (2,) + chi
    flat = rnd(int(np.prod(shp)), scale)
    return flat, flat.reshape(shp[::-1]).transpose(*range(len(shp) - 1, -1, -1)).astype(np.complex128)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def psd(scale=1.0):
    """a BP-like message: Hermitian positive, decaying spectrum (what the kernels multiply with in the engine)"""
    q, _ = np.linalg.qr(rng.standard_normal((32, 32)) + 1j * rng.standard_normal((32, 32)))
    w = np.exp(-np.arange(32) / 4.0)
    return ((q * w) @ q.conj().T * scale).astype(np.complex64).T.reshape(-1).copy()      # M[i, j] at i + 32 j


out = {"route": "f32" if os.environ.get("TNQS_NO_BF16X3") == "1" else "bf16x3", "no_3m": os.environ.get("TNQS_NO_3M") == "1"}
for name, mk in (("iid", lambda: rnd(1024)), ("psd", psd)):
    for (lx, ly) in ((1, 2), (0, 3)):
        fx, tx = site(1e-3); fy, ty = site(1e-3)
        mx = mk(); my = mk()
        Mx = mx.reshape(32, 32).T.astype(np.complex128); My = my.reshape(32, 32).T.astype(np.complex128)
        o = np.zeros_like(fx)
        assert lib.tnqs_dbg_pair_legs(2, z, cchi, lx, ly, P(fx), P(mx), P(my), P(o)) == 0
        ref = np.moveaxis(np.tensordot(tx, Mx, axes=([1 + lx], [0])), -1, 1 + lx)
        ref = np.moveaxis(np.tensordot(ref, My, axes=([1 + ly], [0])), -1, 1 + ly)
        d = o.reshape((2,) + chi, order="F") - ref
        out[f"pair_{name}_{lx}{ly}"] = {"max_over_max": float(np.max(np.abs(d)) / np.max(np.abs(ref))), "rms_over_rms": float(np.sqrt(np.mean(np.abs(d) ** 2) / np.mean(np.abs(ref) ** 2)))}
        oy = np.zeros(1024, np.complex64); ox = np.zeros(1024, np.complex64)
        assert lib.tnqs_dbg_pair_gram2(2, z, cchi, lx, ly, P(fx), P(fy if name == "iid" else fx), P(mx), P(my), P(oy), P(ox)) == 0
        tyy = ty if name == "iid" else tx
        for (absorbed, kept, M, got, tag) in ((lx, ly, Mx, oy, "y"), (ly, lx, My, ox, "x")):
            xm = np.moveaxis(np.tensordot(tx, M, axes=([1 + absorbed], [0])), -1, 1 + absorbed)
            axes = [a for a in range(z + 1) if a != 1 + kept]
            ref = np.tensordot(xm, tyy.conj(), axes=(axes, axes))
            d = got.reshape(32, 32).T - ref
            out[f"gram2_{name}_{lx}{ly}_{tag}"] = {"max_over_max": float(np.max(np.abs(d)) / np.max(np.abs(ref))), "rms_over_rms": float(np.sqrt(np.mean(np.abs(d) ** 2) / np.mean(np.abs(ref) ** 2)))}
print(json.dumps(out, indent=1))
