"""bench.py's no-progress watchdog and error line (CPU): a stalled stage ends the process with ONE parseable JSON line carrying
`error` and `stage` (rank 0), exit code 3 - what a hung collective turns into instead of a launcher that never returns."""
import json
import os
import subprocess
import sys

from _util import ROOT


def test_watchdog_prints_an_error_line_and_exits():
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "import bench\n"
            "wd = bench.Watchdog(1.0, 0, 4)\n"
            "wd.kick('timed steps')\n"
            "time.sleep(30)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["value"] is None and d["n_gpus"] == 4 and d["stage"] == "timed steps" and "no progress" in d["error"]


def test_watchdog_stays_quiet_when_kicked_or_stopped():
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "import bench\n"
            "wd = bench.Watchdog(1.0, 0, 2)\n"
            "for _ in range(5):\n"
            "    time.sleep(0.4); wd.kick('x')\n"
            "wd.stop(); time.sleep(2.0); print('done')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "done", (p.returncode, p.stdout, p.stderr[-500:])
