"""bench.py --gpus N must actually bring up N ranks (the driver's command form is ``python bench.py --gpus N ...``, with or
without torchrun in front).  No GPU here: ``--dryrun-cpu`` swaps in the emulated ops on a tiny model and the gloo backend;
what is under test is the launcher, the rendezvous, the timing protocol and the reported record."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e,
                          timeout=timeout, cwd=ROOT)


def _line(proc):
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                     # ONE JSON line on stdout, whatever the ranks print
    return json.loads(lines[0])


def test_gpus2_spawns_two_ranks_inference():
    rec = _line(_bench("--gpus", "2", "--dryrun-cpu", "--steps", "2", "--warmup", "1", "--latent", "16", "--batch", "2"))
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["scaling"] == "weak" and rec["config"]["global_batch"] == 4
    assert "replicas x2" in rec["config"]["parallelism"]
    assert rec["value"] == pytest.approx(2 * 2 / (rec["ms_per_step"] * 2 * 1e-3), rel=1e-3)
    assert rec["data"].startswith("DRYRUN")


def test_gpus2_spawns_two_ranks_pruning_step():
    rec = _line(_bench("--gpus", "2", "--dryrun-cpu", "--config", "train", "--steps", "2", "--warmup", "1", "--batch", "2"))
    assert rec["n_gpus"] == 2 and "configs[3]" in rec["config"]["workload"]
    assert rec["replicas_identical_after_run"] is True
    spans = rec["collectives"]["spans_ms"]
    assert set(spans) == {"all_gather(text, arch)", "all_gather(sinkhorn scores)", "all_reduce(router grads)"}
    assert 0 < rec["collectives"]["share_of_step"] < 1
    assert rec["per_gpu_steps_per_s"] == pytest.approx(rec["value"] / 2, rel=1e-3)


def test_gpus2_finetune_is_one_expert_per_rank_without_collectives():
    """BASELINE configs[4]: rank r trains expert (r + offset) % 8; the ranks only meet in the timing barrier"""
    rec = _line(_bench("--gpus", "2", "--dryrun-cpu", "--config", "finetune", "--steps", "3", "--warmup", "1", "--batch", "2"))
    assert rec["n_gpus"] == 2 and "configs[4]" in rec["config"]["workload"]
    assert [e["expert"] for e in rec["experts"]] == [0, 1] and [e["rank"] for e in rec["experts"]] == [0, 1]
    assert rec["experts"][0]["keep_ratio"] == 0.4 and rec["experts"][1]["keep_ratio"] == 0.45
    assert "no collective" in rec["config"]["parallelism"] and rec["scaling"] == "weak"
    assert rec["value"] == pytest.approx(2 * 3 / (rec["ms_per_step"] * 3 * 1e-3), rel=1e-3)
    one = _line(_bench("--dryrun-cpu", "--config", "finetune", "--steps", "2", "--warmup", "0", "--batch", "2"))
    assert one["n_gpus"] == 1 and one["experts"][0]["expert"] == 3                # N = 1: expert 3
    off = _line(_bench("--gpus", "2", "--dryrun-cpu", "--config", "finetune", "--steps", "1", "--warmup", "0", "--expert-offset", "6"))
    assert [e["expert"] for e in off["experts"]] == [6, 7]


def test_single_rank_default_and_world_size_mismatch():
    rec = _line(_bench("--dryrun-cpu", "--steps", "1", "--warmup", "0", "--latent", "16", "--batch", "1"))
    assert rec["n_gpus"] == 1
    bad = _bench("--gpus", "2", "--dryrun-cpu", "--steps", "1", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
