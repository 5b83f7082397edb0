"""The N = 1 bench line on a small corpus (what the driver runs as `python bench.py --gpus 1 --steps K --warmup W`): the contract's
fields, `value` on HBM-resident batches with the PCIe-inclusive form timed beside it, parity fields, the side blocks' own checks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra, env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_single_gpu_contract_fields():
    d = _run(["--workload", "small", "--steps", "6", "--warmup", "2", "--settle-ms", "20", "--pipelined-steps", "6", "--cpu-seconds", "2",
              "--cpu-max-queries", "4", "--latency-tail", "300", "--coalesced-calls", "20"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "queries_per_sec" and d["unit"] == "queries/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["value"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert abs(d["value"] - d["config"]["batch"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    # `value`: the batch is resident in HBM when a step starts; the PCIe-inclusive form is timed beside it, never as `value`
    assert d["config"]["queries_start_in"] == "hbm" and "workload" in d["config"]
    pi = d["pcie_inclusive_serial"]
    assert pi["ms_per_step"] > 0 and pi["same_results_as_value_step"] is True
    assert d["hbm_resident_results_in_hbm"]["same_results_as_value_step"] is True
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and 0.0 < r["frac"] < 1.0 and r["achieved"] > 0 and r["peak"] > 0 and "traffic" in r
    assert r["avg_launch_ms"] > 0 and r["algorithmic_flops_per_launch"] == 2.0 * d["config"]["batch"] * r["rows_per_launch"] * d["config"]["dim"]
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] == "port" and c["sample"] and c["inputs_same_as_cpu_generator"] is True
    assert d["recall_at_10"] == d["ordered_match_rate"] == d["score_bit_exact_rate"] == 1.0
    ex = d["exact_check"]
    assert ex["queries"] == d["config"]["batch"] and ex["recall_at_10"] == ex["ordered_match_rate"] == ex["score_bit_exact_rate"] == 1.0
    assert d["pipelined_host"]["same_results_as_serial_step"] is True and d["concurrent_callers"]["same_results_as_serial_step"] is True
    assert d["coalesced_callers"]["same_results_as_lone_calls"] is True
    assert d["latency"]["nq1"]["same_results"] is True and d["latency"]["nq1_tail"]["calls"] == 300
    assert d["pipeline"]["fallback_queries"] == 0


def test_bench_line_host_resident_queries_form():
    """--queries host: SURVEY.md 8(d)'s step (the batch crosses PCIe inside it) is `value`, the HBM-resident form rides along."""
    d = _run(["--workload", "small", "--queries", "host", "--steps", "5", "--warmup", "2", "--settle-ms", "20", "--pipelined-steps", "0",
              "--cpu-seconds", "0", "--latency", "0", "--coalesced-threads", "0", "--callers", "0"])
    assert d["config"]["queries_start_in"] == "pinned host memory" and "pcie_inclusive_serial" not in d
    assert d["hbm_resident_serial"]["same_results_as_value_step"] is True and d["hbm_resident_serial"]["ms_per_step"] > 0
    assert d["cpu_baseline"] is None or "cpu_baseline" in d
    ex = d["exact_check"]
    assert ex["recall_at_10"] == ex["ordered_match_rate"] == ex["score_bit_exact_rate"] == 1.0
