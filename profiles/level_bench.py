"""One BP level of degree-4 chi = 32 sites (pair product on two legs, both messages through the other two) on device-resident tensors, the sites
processed in groups (tnqs_dbg_bench_plane which = 4, TNQS_DBG_GROUP): does a group small enough for the Infinity Cache pay?
    python profiles/level_bench.py [nsites] [reps] [group sizes ...]"""
import ctypes as C, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    lib = C.CDLL(os.path.join(here, "..", "tensornetworkquantumsimulator.jl_amd", "libtnqs_hip.so"))
    lib.tnqs_dbg_bench_plane.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    ms = C.c_double(0)
    rc = lib.tnqs_dbg_bench_plane(4, int(sys.argv[2]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[3]), C.byref(ms))
    print(f"rc {rc}  {ms.value:8.3f} ms per level of {sys.argv[2]} sites = {ms.value / int(sys.argv[2]) * 100:.3f} ms per 100 sites")
    sys.exit(0)
nsites = sys.argv[1] if len(sys.argv) > 1 else "96"
reps = sys.argv[2] if len(sys.argv) > 2 else "3"
groups = sys.argv[3:] or ["0", "16", "8", "6", "4", "3", "2"]
for (lx, ly) in ((0, 3), (1, 2)):
    for gsz in groups:
        out = subprocess.run([sys.executable, __file__, "--one", nsites, reps, str(lx), str(ly)], env=dict(os.environ, TNQS_DBG_GROUP=gsz), capture_output=True, text=True)
        print(f"pair legs ({lx},{ly}) group {gsz:>3s}: {out.stdout.strip()}  {out.stderr.strip().splitlines()[-1] if out.stderr.strip() else ''}", flush=True)
