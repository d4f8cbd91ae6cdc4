"""bench.py's launch contract: `python bench.py --gpus N` with no launcher around it (the driver's command shape) must become N ranks and
print "n_gpus": N -- or fail loudly; it must never run one rank and print n_gpus 1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


def test_gpus_flag_without_enough_devices_fails_loudly():
    """CPU box (or any box with fewer than 64 GPUs): the request cannot be served, the script says so and exits non-zero without a JSON line"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 64 visible GPUs" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_world_size_must_agree_with_the_gpus_flag():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_gpus_2_spawns_two_ranks_and_reports_them():
    """functional test of the launch path on a one-GPU box: gloo rendezvous, the two ranks share the device (callback transport).  With
    two real GPUs the same command without TNQS_BENCH_BACKEND runs one rank per GPU on the library's RCCL transport."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--L", "4", "--chi", "4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=clean_env(TNQS_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["transport"]["nranks"] == 2
    assert out["config"]["transport"]["allgathers_per_step"] > 0 and out["value"] > 0


@pytest.mark.gpu
def test_single_gpu_line_carries_roofline_and_no_transport():
    r = subprocess.run([sys.executable, BENCH, "--L", "4", "--chi", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["transport"] is None and "kernel_classes" in out


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,L", [("c4", "3"), ("c5", "4")])
def test_big_configurations_run_sharded_over_gloo(cfg, L):
    """bench.py --config c4 / c5 (BASELINE configs[3] / [4]: the 8-GPU shapes) on lattices that fit one GPU, two ranks sharing the device over
    gloo: the state is generated on the device rank by rank (tnqs_set_site_random), the line names the BASELINE entry and carries the memory
    estimate, both flop counts and the sweep order."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", cfg, "--L", L, "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=clean_env(TNQS_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = out["config"]
    assert out["n_gpus"] == 2 and c["baseline_config"] == cfg and ("configs[3]" if cfg == "c4" else "configs[4]") in c["workload"]
    assert c["state_init"].startswith("on device") and c["memory"]["site_tensor_GiB_this_rank"] > 0 and "bp_order" in c
    assert out["flop_counts"]["executed_algorithm_TFLOP_per_step"] > 0 and out["flop_counts"]["reference_order_TFLOP_per_step"] > 0
    assert out["value"] > 0 and out["roofline"] is not None


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,L,chi", [("c2", "8", "8"), ("c4", "4", "4")])
def test_eight_ranks_over_gloo(cfg, L, chi):
    """round-5 verdict item 7: the driver's 8-GPU command shape, `python bench.py --gpus 8`, run for real -- eight ranks sharing this box's one GPU over gloo (callback
    transport; with eight GPUs the same command runs one rank per GPU on the library's RCCL transport): launch, work-balanced partition, every exchange of a
    layer, the max-over-ranks timing and the JSON line.  The bytes the library counted per step must be what dist.exchange_plan predicts from the slot rules --
    the restatement the CPU suite sizes the exchange buffer of the full-size configurations with (tests/test_sharding_cpu.py)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--config", cfg, "--L", L, "--chi", chi, "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=clean_env(TNQS_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    tr = out["config"]["transport"]
    assert out["n_gpus"] == 8 and tr["nranks"] == 8 and len(tr["partition"]["vertices"]) == 8 and min(tr["partition"]["vertices"]) > 0
    assert out["value"] > 0 and out["config"]["bp_sweeps_per_step"][0] >= out["config"]["bp_updates_per_step"][0]
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tnqs_amd as tn
    from test_sharding_cpu import _default_levels
    Li, ci = int(L), int(chi)
    g = tn.named_grid((Li, Li, Li), periodic=True) if cfg == "c4" else tn.named_grid((Li, Li))
    groups = tn.edge_color(g) if cfg == "c4" else tn.edge_color(g, 4)
    owner = tn.partition_vertices(g.nv(), 8, tn.dist.site_weights(g, ci))
    nup, nsw = out["config"]["bp_updates_per_step"][0], out["config"]["bp_sweeps_per_step"][0]
    plan = tn.dist.exchange_plan(g, owner, ci, groups, _default_levels(g), bp_updates_per_layer=1, sweeps_per_update=nsw)       # (sweeps of all updates of the step)
    print(cfg, "library:", tr["allgathers_per_step"], "all-gathers,", tr["MB_gathered_per_step"], "MB per step; plan:", plan["exchanges_per_layer"], plan["bytes_gathered_per_layer"] / 1e6, "updates", nup, "sweeps", nsw)
    # both counters average the timed step and the warm-up step, which may have swept more often: the plan of the TIMED step is a lower bound within a sweep's worth
    per_sweep = sum(1 for _ in _default_levels(g))
    assert abs(tr["allgathers_per_step"] - plan["exchanges_per_layer"]) <= 8 * per_sweep
    assert abs(tr["MB_gathered_per_step"] - plan["bytes_gathered_per_layer"] / 1e6) <= 0.25 * plan["bytes_gathered_per_layer"] / 1e6


def test_memory_estimate_refuses_what_cannot_fit():
    """--config c4 at full size on ONE rank is 250 GiB of site tensors: refused before anything is allocated (on a CPU box the script stops earlier,
    at the missing device -- either way no JSON line and a non-zero exit)"""
    r = subprocess.run([sys.executable, BENCH, "--config", "c4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout
