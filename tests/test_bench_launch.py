"""bench.py --gpus N without a launcher starts N ranks itself (CPU: the launch logic only, no GPU is touched)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_gpus_n_spawns_n_fresh_ranks_and_relays_rank0():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--spawn-check"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip())          # stdout carries rank 0's line only
    assert line == {"rank": 0, "world": 3, "local_rank": 0, "master": "127.0.0.1", "torch_loaded_in_parent": False}
    others = sorted(json.loads(l)["rank"] for l in r.stderr.splitlines() if l.startswith("{"))
    assert others == [1, 2]


def test_rank_count_and_gpus_must_agree():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "must agree" in r.stderr


def test_parent_of_the_ranks_never_imports_torch():
    """The spawning parent must not have initialised the GPU runtime (a process that has cannot safely start GPU children):
    bench.py imports torch only inside main(), after the spawn decision."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    top = [n for n in ast.parse(src).body if isinstance(n, (ast.Import, ast.ImportFrom))]
    names = {a.name.split(".")[0] for n in top if isinstance(n, ast.Import) for a in n.names} | {(n.module or "").split(".")[0] for n in top if isinstance(n, ast.ImportFrom)}
    assert "torch" not in names and "lmat_amd" not in names
    body = src[src.index("def main()"):]
    assert body.index("spawn_ranks(args.gpus)") < body.index("import torch")
