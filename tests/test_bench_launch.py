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


def test_a_dead_rank_takes_the_others_down():
    """One rank exits non-zero while its siblings would wait (in a collective) for ever: the parent stops them and fails."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--spawn-check"],
                       env=dict(_env(), LMAT_SPAWN_CHECK_FAIL="1"), capture_output=True, text=True, timeout=100)
    assert r.returncode == 3 and time.monotonic() - t0 < 60, (r.returncode, r.stderr)


def test_step_plan_weak_and_strong():
    """--total-reads N is strong scaling: N reads over the ranks and the steps; --batch is weak scaling."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(total_reads=0, batch=8_000_000, launch_reads=8_000_000, steps=20)
    assert bench.plan_steps(a, 8) is False and a.batch == 8_000_000 and a.launch_reads == 8_000_000
    a = argparse.Namespace(total_reads=0, batch=30000, launch_reads=10000, steps=2)
    assert bench.plan_steps(a, 2) is False and (a.batch, a.launch_reads) == (30000, 10000)
    for world in (1, 2, 4, 8):   # BASELINE config 4: 50 M reads in all
        a = argparse.Namespace(total_reads=50_000_000, batch=8_000_000, launch_reads=8_000_000, steps=20)
        assert bench.plan_steps(a, world) is True
        total = a.batch * a.steps * world
        assert 50_000_000 <= total < 50_000_000 + 20 * world * 2 and a.batch % a.launch_reads == 0
    a = argparse.Namespace(total_reads=400_000_000, batch=1, launch_reads=8_000_000, steps=10)   # a step of several launches
    assert bench.plan_steps(a, 2) is True and 20_000_000 <= a.batch <= 20_000_002 and a.batch == 3 * a.launch_reads
