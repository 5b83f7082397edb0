"""bench.py --gpus N without a launcher must start N ranks itself (VERDICT r1 'Missing' 2): the driver may run
`python3 bench.py --gpus N` plainly. --spawn-check makes every rank report its place and exit before touching a GPU."""
import json
import os
import subprocess
import sys

from _util import ROOT


def test_bench_self_spawns_ranks_without_a_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert all(l["world"] == 2 and l["master"] == "127.0.0.1" for l in lines)


def test_bench_under_launcher_uses_the_given_world():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["world"] == 1
