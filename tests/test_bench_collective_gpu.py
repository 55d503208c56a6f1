"""bench.py's N > 1 code path on ONE MI355X: `init_process_group("nccl")` before anything touches the GPU, the RCCL
all-gather of the per-trajectory summaries inside the timed region, the barriers and the max-over-ranks reduction -- run
as a FRESH child process (the process group must exist before the first GPU call; a process that has touched the GPU is
never re-exec'ed), once through `--force-collective` and once under `torch.distributed.run --nproc-per-node 1`, for every
config.  No scaling curve comes out of this (one rank): it proves the code the driver's N = 2/4/8 runs execute has run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(proc):
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-4000:])
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout
    return json.loads(lines[0])


SMALL = {"kalman4": ["--batch", "4096", "--T", "400"], "gsf32": ["--batch", "512", "--T", "100", "--chunk", "50"],
         "bpf4096": ["--batch", "32", "--T", "20"], "kalman64": ["--batch", "512", "--T", "200"]}


@pytest.mark.parametrize("config", sorted(SMALL))
def test_bench_under_torchrun_one_rank(config):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--config", config] + SMALL[config]
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert d["n_gpus"] == 1 and d["scaling"] in ("weak", "strong") and d["steps"] == 2 and d["warmup"] == 1
    assert "RCCL all-gather" in d["config"]["parallelism"]
    assert d["roofline"]["bound"] in ("hbm", "mfma", "valu") and d["roofline"]["kernel_ms"] > 0
    assert d["value"] > 0 and d["unit"] == "timesteps/s" and d["finite_frac"] > 0.5
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.001          # the kernels' HIP-event time sits inside the step


def test_bench_force_collective_flag():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-collective", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + SMALL["kalman4"]
    d = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert d["n_gpus"] == 1 and "RCCL all-gather" in d["config"]["parallelism"] and d["roofline"]["frac"] > 0
    # ... and without the flag the plain single-GPU path (no process group) prints the same schema
    cmd.remove("--force-collective")
    d0 = _line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert "RCCL" not in d0["config"]["parallelism"] and set(d0) == set(d)
