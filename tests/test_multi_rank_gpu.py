"""Rank equality of REAL generations (`-m gpu`): two ranks (one process each, as `bench.py --gpus 2` starts them) generate
their shard of the images through the full-size SD1.5 fused loop, and rank 0 regenerates every rank's images itself - the
sha256 of the final latents must be equal, i.e. a rank computes exactly what a single process computes for the same image
(image i -> rank i mod G, no per-step collective; DESIGN.md section 7: every kernel of the step is
bit-reproducible and none is chosen per process).

On the one-GPU test box the two ranks share the card and talk over gloo (`DSC_DIST_BACKEND=gloo`; RCCL refuses two ranks on
one device), so this checks the launch path, the sharding, the broadcast and the arithmetic - not RCCL.  The ranks are
CHILD processes of the test (subprocess -> torch.distributed.run -> rank processes, started by bench.py's `launch_ranks`
before that launcher process has touched the GPU); the pytest process itself only waits for them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_produce_the_single_process_latents():
    env = dict(os.environ, DSC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--in-flight", "1",
                        "--no-cpu-baseline", "--no-batched-roofline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == 2 and len(cfg["per_rank_images_per_s"]) == 2 and len(cfg["devices"]) == 2
    assert cfg["ranks_equal_single_process"] is True                       # sha256 of every rank's final latents == rank 0's own run
    assert cfg["outputs_finite"] is True
    assert "gloo" in cfg["dist_backend"] and "NOT RCCL" in cfg["dist_backend"]      # the line says what it is
