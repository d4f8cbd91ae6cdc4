"""CPU: the COMPILED CPU restatement (oracle/cpu_port.cpp: bench.py's `cpu_baseline`) against the numpy oracle (oracle/tnqs_oracle.py), which follows the
reference line by line and is what every parity test compares the device with.  Both are test / bench infrastructure.  Compared: BP messages after an
update with an explicit common sweep order (elementwise: same tensors, same order), then TFIM layers -- bond dimensions, truncation errors, and <Z> on every site through the oracle's own expect on the downloaded state (gauge-invariant quantities only: the two sides call
different LAPACK drivers, so singular vectors carry different phases)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tnqs_oracle as o  # noqa: E402
import cpu_port  # noqa: E402


def to_oracle(net, g):
    tns = o.TensorNetworkState(g, {v: net.tensor(v) for v in g.vertices})
    b = o.BeliefPropagationCache(tns, edge_sequence=[])
    for (a, c) in g.edges:
        b.messages[(a, c)] = net.message(a, c); b.messages[(c, a)] = net.message(c, a)
    return b


@pytest.mark.parametrize("make,chi", [(lambda: o.named_grid((3, 3)), 4), (lambda: o.named_grid((4, 3)), 6), (lambda: o.heavy_hexagonal_lattice(1, 1), 4),
                                      (lambda: o.named_grid((2, 2, 2)), 3)])
def test_compiled_port_matches_the_numpy_oracle(make, chi):
    cpu_port.build()
    g = make()
    psi = o.random_state(np.complex64, g, chi, seed=7)
    groups = o.edge_color(g)
    seq = []
    for grp in groups:
        seq += list(grp) + [(b, a) for (a, b) in grp]
    bo = o.update(o.BeliefPropagationCache(psi, edge_sequence=seq), maxiter=8, tolerance=None)
    net = cpu_port.CpuNet.from_oracle(o.BeliefPropagationCache(psi, edge_sequence=seq))
    net.update(seq, maxiter=8, tolerance=None)
    for (a, b) in seq:                                    # same tensors, same order, same number of sweeps: elementwise
        assert np.max(np.abs(net.message(a, b) - bo.message((a, b)))) < 2e-5 * np.max(np.abs(bo.message((a, b)))), (a, b)
    one = [("Rx", [v], 0.3) for v in g.vertices]
    cg = [[("Rzz", [a, b], 0.4) for (a, b) in grp] for grp in groups]
    layer = one + [gt for grp in cg for gt in grp]
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    zop = np.diag([1.0, -1.0]).astype(complex)
    for it in range(2):
        bo, eo = o.apply_gates(layer, bo, apply_kwargs=kw, bp_update_kwargs=dict(maxiter=30, tolerance=1e-7, edge_sequence=seq))
        ec, sweeps = cpu_port.apply_layer(net, one, cg, seq, kw, maxiter=30, tolerance=1e-7)
        eo = np.array(eo)[len(one):]                      # (the oracle lists an error of 0 for every one-site gate first)
        assert [net.bond_dim(a, b) for (a, b) in g.edges] == [bo.tns.bond_dim(a, b) for (a, b) in g.edges], it
        assert np.all(np.abs(ec - eo) < 2e-3 * np.maximum(ec, eo) + 3e-7), (it, float(np.max(np.abs(ec - eo))))
        bc = to_oracle(net, g)
        zc = np.array([o.expect_1site(bc, zop, v).real for v in g.vertices]); zo = np.array([o.expect_1site(bo, zop, v).real for v in g.vertices])
        assert np.max(np.abs(zc - zo)) < 1e-5, (it, float(np.max(np.abs(zc - zo))))


def test_compiled_port_is_not_imported_by_the_product():
    import subprocess
    code = "import sys; sys.path.insert(0, %r); import tnqs_amd; assert 'cpu_port' not in sys.modules and 'tnqs_oracle' not in sys.modules" % ROOT
    assert subprocess.run([sys.executable, "-c", code], capture_output=True).returncode == 0


def test_cpu_budget_reads_the_cgroup_quota(tmp_path, monkeypatch):
    """the GPU boxes show 256 hardware threads and run under cpu.max = 16 CPUs: the team is sized to the quota, not to what /proc/cpuinfo lists"""
    import builtins
    real_open = builtins.open

    def fake(text):
        f = tmp_path / "cpu.max"; f.write_text(text)

        def _open(path, *a, **k):
            return real_open(str(f), *a, **k) if path == "/sys/fs/cgroup/cpu.max" else real_open(path, *a, **k)
        return _open
    monkeypatch.setattr(cpu_port.os, "sched_getaffinity", lambda _pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert cpu_port.cpu_budget() == (16, 16.0)
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert cpu_port.cpu_budget() == (128, None)                     # no quota: the physical cores (SMT-2)
    monkeypatch.setattr(builtins, "open", fake("50000 100000\n"))
    assert cpu_port.cpu_budget()[0] == 1                            # half a CPU still runs one thread
