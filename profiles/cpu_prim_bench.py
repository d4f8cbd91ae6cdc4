"""Host-side primitive rates of the CPU baseline (oracle/cpu_layer.py) under the concurrency it runs at: N threads, one BLAS thread each,
every thread its own chi = 32 site tensor.  python profiles/cpu_prim_bench.py [nthreads]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import tnqs_oracle as o
from threadpoolctl import threadpool_limits
from concurrent.futures import ThreadPoolExecutor

class S:
    _max_workers = 1
    @staticmethod
    def map(fn, it): return map(fn, it)

o._BIG = 1 << 12
o._POOL = S()
nt = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, min(64, (os.cpu_count() or 2) // 2))
rng = np.random.default_rng(0)
chi = 32
shape = (2, chi, chi, chi, chi)
m = (rng.standard_normal((chi, chi), dtype=np.float32) + 1j).astype(np.complex64)
gf = 8.0 * 2 * chi ** 5 / 1e9
with threadpool_limits(limits=1), ThreadPoolExecutor(nt) as pool:
    ts = list(pool.map(lambda i: (np.random.default_rng(i).standard_normal(shape, dtype=np.float32) + 1j).astype(np.complex64), range(nt)))   # first touch by a pool thread
    def run(name, fn, flops):
        list(pool.map(fn, ts))
        t0 = time.perf_counter(); list(pool.map(fn, ts)); dt = time.perf_counter() - t0
        print(f"{name:22s} {nt} threads  {dt*1e3:8.1f} ms  {flops*nt/dt:8.1f} GF/s total", flush=True)
    for ax in (1, 2, 3, 4):
        run(f"absorb axis {ax}", lambda t, ax=ax: o._absorb(t, ax, m), gf)
    for ax in (1, 2, 3, 4):
        run(f"gram axis {ax}", lambda t, ax=ax: o._gram(t, t, ax), gf)
    run("qr 65536x64 lapack", lambda t: np.linalg.qr(t.reshape(-1, 2 * chi)[:65536], mode="reduced"), 8.0 * 65536 * 64 * 64 * 2 / 1e9)
    for blk in (2048, 4096, 8192, 16384):
        o._QR_BLOCK = blk
        run(f"qr 65536x64 blocked {blk}", lambda t: o._qr_thin(t.reshape(-1, 2 * chi)[:65536]), 8.0 * 65536 * 64 * 64 * 2 / 1e9)
    run("copy 16 MB", lambda t: t.copy(), 0.0)
    run("astype c128", lambda t: t.astype(np.complex128), 0.0)
