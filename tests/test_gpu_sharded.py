"""GPU: the vertex-sharded (2-rank) path of the library against the unsharded one.  Two processes share GPU 0 and
exchange through torch.distributed's gloo backend (host-staged); in production the same callback runs on RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dt,nranks", [("c128", 2), ("c64", 2), ("chi32", 2), ("illc128", 2), ("c128", 4), ("illc128", 4),
                                       ("z6chi16", 2), ("z4chi64", 2)])
def test_sharded_apply_gates_matches_single_rank(tmp_path, dt, nranks):
    """2 and 4 ranks (with 4, ranks own three or four vertices each and sit out whole colour batches -- every collective must still be
    issued by all of them).  A one-off run with 8 ranks of all four cases gave the same deviations."""
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29533 + nranks), os.path.join(ROOT, "tests", "sharded_worker.py"), out, dt]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    # "illc128": ComplexF64, cutoff = 1e-14 -- agreement to 1e-11 needs the second factorisation pass on both sides (1e-9 with a single pass)
    tol = 1e-9 if dt == "c128" else (1e-11 if dt == "illc128" else (2e-4 if dt == "c64" else 5e-4))      # chi32, z6chi16, z4chi64: 5e-4
    assert np.array_equal(z["dims_sh"], z["dims_un"])
    assert np.max(np.abs(z["errs_sh"] - z["errs_un"])) < (1e-10 if dt in ("c128", "illc128") else 1e-5)
    print(dt, "max |<Z>_sharded - <Z>_single| =", np.max(np.abs(z["ez_sh"] - z["ez_un"])), " spectra:", np.max(np.abs(z["sp_sh"] - z["sp_un"])))
    assert np.max(np.abs(z["ez_sh"] - z["ez_un"])) < tol
    assert np.max(np.abs(z["sp_sh"] - z["sp_un"])) < tol
    assert int(z["n_exchanges"]) > 0
