"""CPU (gloo, world_size 2): the vertex-sharded protocol -- what is computed where and what is exchanged -- gives the
same truncation errors, bond dimensions and <Z> as the serial algorithm; plus the host-side partition helper."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

import tnqs_amd as tn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_vertices():
    for nv, w in [(400, 8), (25, 4), (7, 2), (3, 3), (5, 1)]:
        own = tn.partition_vertices(nv, w)
        assert len(own) == nv and sorted(set(own)) == list(range(w))
        counts = np.bincount(own, minlength=w)
        assert counts.max() - counts.min() <= 1
        assert own == sorted(own)
    with pytest.raises(ValueError):
        tn.partition_vertices(0, 2)
    assert tn.dist.exchange_bytes_needed(32, 2, 760, 400, 8) < 200 << 20


def test_two_rank_protocol_matches_serial_oracle(tmp_path):
    out = str(tmp_path / "sim.pkl")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "sharding_sim_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = pickle.load(open(out, "rb"))
    assert z["dims"] == z["odims"]
    assert np.max(np.abs(z["errs"] - z["oerrs"])) < 1e-12
    assert np.max(np.abs(z["ez"] - z["oez"])) < 1e-10
    assert sorted(set(z["owner"])) == [0, 1]
