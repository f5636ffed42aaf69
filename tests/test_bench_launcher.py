"""`python bench.py --gpus N` outside torch.distributed.run must start N ranks itself (it used to run ONE rank and print n_gpus: 1);
under torch.distributed.run (the driver's multi-GPU form) it must not re-spawn.  CPU only: the `stub` workload does one gloo all_gather per step."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "3", "--warmup", "1",
                        "--master-port", str(_free_port())], capture_output=True, text=True, env=_env(), timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak"


def test_bench_under_torchrun_does_not_respawn():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, env=_env(), timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2


def test_bench_single_rank_and_mismatch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub", "--steps", "2", "--warmup", "0"], capture_output=True,
                       text=True, env=_env(), timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 1
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "stub"], capture_output=True, text=True, env=env,
                       timeout=240, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
