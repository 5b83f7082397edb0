"""The group-commit layer (codegraph-rust_amd/csrc/coalesce.h) is host-only C++: tests/c_client/coalesce_host.cpp drives it with a
fake device - many caller threads, mixed nq and k in one batch, poisoned requests that must fail their own caller only - and once
more under ThreadSanitizer (sanitizers run on the CPU build only). No GPU needed."""
import json
import os
import subprocess

import pytest

from _util import ROOT

SRC = os.path.join(ROOT, "tests", "c_client", "coalesce_host.cpp")


def _build(tmp_path, name, extra=()):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-pthread", *extra, SRC, "-o", exe])
    return exe


@pytest.mark.parametrize("threads,calls,leaders,window,kclasses",
                         [(1, 100, 2, 0, 1), (16, 150, 2, 0, 1), (64, 60, 2, 0, 1), (64, 60, 1, 250, 1), (32, 80, 3, 40, 1),
                          (24, 150, 1, 250, 3), (64, 60, 1, 0, 4), (9, 300, 2, 30, 2), (3, 400, 1, 250, 2)])
def test_every_caller_gets_its_own_answers(tmp_path, threads, calls, leaders, window, kclasses):
    """(a hang - a member never woken, a lost hand-over - ends in the timeout)"""
    exe = _build(tmp_path, "coalesce_host")
    p = subprocess.run([exe, str(threads), str(calls), str(leaders), str(window), "150", str(kclasses)], capture_output=True, text=True,
                       timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    d = json.loads(p.stdout)
    assert d["bad"] == 0 and d["calls"] == threads * calls
    assert d["max_batch_queries"] <= 64
    if threads == 1:   # a lone caller never waits and never stages: every call runs alone
        assert d["lone_calls"] == calls and d["batches"] == 0
    else:
        assert d["batches"] > 0 and d["batched_requests"] > d["batches"]
        assert d["poisoned_ok"] > 0   # (each failed its own caller only: bad == 0 covers the neighbours)


def test_thread_sanitizer_clean(tmp_path):
    exe = _build(tmp_path, "coalesce_host_tsan", extra=("-fsanitize=thread",))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    for args in (("16", "60", "2", "20", "100"), ("32", "40", "1", "0", "100"), ("24", "60", "1", "250", "60", "3")):
        p = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0 and "ThreadSanitizer" not in p.stderr, p.stdout + p.stderr[-3000:]
