"""bench.py's launcher contract, without a GPU (`--launch-check`: ranks rendezvous over gloo and count themselves):
`python bench.py --gpus N` with no launcher in the environment must start N ranks itself - a silent single-rank run that prints
n_gpus 1 is what a scaling run must never get - and a WORLD_SIZE that disagrees with --gpus must fail."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_plain_invocation_spawns_the_ranks_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2


def test_world_size_mismatch_is_an_error():
    env = _clean_env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_single_rank_launch_check():
    r = subprocess.run([sys.executable, BENCH, "--launch-check"], capture_output=True, text=True, env=_clean_env(), timeout=120)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_ranks_seen"] == 1


def test_extra_legs_guard_fails_the_run_on_hang_and_error():
    """ADVICE r2 / VERDICT r2 weak 8(d): a hung or failed collective of the extra legs must reach the driver as a non-zero exit,
    after the line (with `extra_legs_status`) has been printed from a consistent snapshot."""
    for how, status in (("hang", "hang"), ("error", "error")):
        r = subprocess.run([sys.executable, BENCH, "--launch-check", "--selftest-legs", how], capture_output=True, text=True,
                           env=_clean_env(), timeout=120)
        assert r.returncode == 3, (how, r.returncode, r.stdout, r.stderr)
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert out["extra_legs_status"] == status and out["extra_legs"]["first"] == {"ok": True}
    r = subprocess.run([sys.executable, BENCH, "--launch-check", "--selftest-legs", "ok"], capture_output=True, text=True,
                       env=_clean_env(), timeout=120)
    assert r.returncode == 0
    assert json.loads(r.stdout.strip().splitlines()[-1])["extra_legs_status"] == "ok"


def test_one_gpu_rehearsal_is_refused_for_modes_that_open_rccl_groups():
    """--rehearse-one-gpu puts every rank on cuda:0; the sharded modes and the extra legs would bootstrap RCCL groups with two
    ranks on one device. They must be refused before any rank starts."""
    for extra in (["--mode", "sp"], ["--mode", "cfg-pair"], ["--mode", "vae-tiles"], ["--extra-legs"]):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-one-gpu"] + extra, capture_output=True, text=True,
                           env=_clean_env(), timeout=120)
        assert r.returncode != 0 and "replica path only" in r.stderr, (extra, r.returncode, r.stderr[-300:])
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_two_rank_replica_line_rehearsed_on_one_gpu():
    """The N > 1 code of bench.py (launcher, rank count by all-reduce, context broadcast, barrier-bracketed timing, MAX over ranks,
    per-rank times) executed with two ranks that share the box's one GPU over gloo (--rehearse-one-gpu). Not a scaling number - the
    line says so - but every statement a real 2-GPU run executes outside RCCL has run before the driver's scaling run does."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-one-gpu", "--steps", "2", "--warmup", "1", "--no-vae", "--no-aux",
                        "--no-prof"], capture_output=True, text=True, env=_clean_env(), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["scaling"] == "weak" and "rehearsal" in out
    assert len(out["per_rank_ms_per_step"]) == 2 and out["value"] > 0
    assert abs(out["value"] - 2 * 1e3 / out["ms_per_step"]) < 0.05 * out["value"]   # whole-job steps/s = ranks x steps / max time


def test_cpu_baseline_legs_run_on_a_small_case():
    """bench.py's cpu_baseline legs (the oracle timed on the host): the whole-unit path and the whole-blocks fallback, on a case small enough
    for the CPU suite - the driver's bench line depends on these helpers never raising."""
    sys.path.insert(0, ROOT)
    import bench

    r = bench.cpu_baseline(T=16, S=8, F=1, H=4, W=4, budget_s=1e9)          # whole step
    assert r["extrapolated"] is False and r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1 and "ONE whole denoise step" in r["sample"]
    r2 = bench.cpu_baseline(T=16, S=8, F=1, H=4, W=4, budget_s=1e-9)        # a host too slow for the budget: whole blocks, said so
    assert r2["extrapolated"] is True and r2["value"] > 0 and "scaled x48/" in r2["sample"]
    v = bench.cpu_baseline_vae(1, 2, 2, budget_s=1e9)
    assert v["extrapolated"] is False and v["decode_ms"] > 0 and "ONE whole oracle.decode_video" in v["sample"]
