"""`python bench.py --gpus N` must start its own N ranks: the driver runs exactly that command for the
scaling curve.  Here, without a GPU, `--dry-run` replaces the kernel by a stub on CPU tensors over gloo -- everything
around the kernel (self-launch through torch.distributed.run before any GPU call, rendezvous on 127.0.0.1, disjoint lane
seeds, one all-gather per group through oak_amd.dist.gather_round, the row check, max-over-ranks clock, ONE JSON line
from rank 0) is the code the GPU run uses."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    return p


def test_gpus_2_self_launches_two_ranks_dry():
    p = _run(["--gpus", "2", "--steps", "5", "--warmup", "0", "--group", "2", "--batch", "512", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["dry_run"] is True and rec["n_gpus"] == 2 and rec["ranks_seen"] == 2
    assert rec["scaling"] == "weak" and rec["steps"] == 5 and rec["config"]["groups"] == 3
    s0, s1 = rec["config"]["first_lane_seed_per_rank"]
    assert s1 - s0 == rec["config"]["batch_per_gpu"] == 512          # contiguous, disjoint lane blocks
    assert "torch.distributed.run" in p.stderr                         # the launch is announced on stderr


def test_a_failing_rank_fails_the_command():
    # the launcher's exit code is the command's: a rank that dies must not look like a finished benchmark
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])          # no --dry-run and no GPU here: every rank exits non-zero
    if p.returncode == 0:                                                # (on a GPU box with >= 2 GPUs this really runs)
        assert any(l.startswith("{") for l in p.stdout.splitlines())
    else:
        assert not any(l.startswith("{") for l in p.stdout.splitlines())


def test_world_size_mismatch_is_refused():
    p = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


import pytest


@pytest.mark.gpu
def test_gpus_2_rehearsal_on_one_gpu_runs_the_real_exchange_path():
    """BENCH_REHEARSAL=1: `python bench.py --gpus 2` self-launches two ranks that SHARE GPU 0 over gloo -- the real kernels and
    the real N > 1 code path of every record (disjoint lane blocks, one gather per group on its own stream, the row check of
    the gather, sums over ranks, config4's root split with its all-gather of per-root means) on a one-GPU box."""
    p = _run(["--gpus", "2", "--steps", "8", "--warmup", "2", "--group", "4", "--batch", "16384", "--no-cpu-baseline"], {"BENCH_REHEARSAL": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and "rehearsal" in rec
    assert 90 < rec["config"]["mean_turn_steps_per_playout"] < 110          # 2 x 8 x 16,384 playouts, all counted
    for sub in ("leaf", "config3", "config4", "config5"):
        assert sub in rec and "error" not in rec[sub], (sub, rec.get(sub))
    assert rec["config4"]["scaling"] == "strong" and rec["config4"]["config"]["roots_per_gpu"] == 128
    assert rec["config4"]["ranks_seen"] == 2 and 0.3 < rec["config4"]["config"]["mean_root_value"] < 0.7
