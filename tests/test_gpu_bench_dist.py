"""bench.py's world > 1 PROGRAM executed on a 1-GPU box (VERDICT r4 'Next' 2): `--dist-backend gloo` runs the same rank program
the driver starts for its 2 / 4 / 8-GPU scaling runs - shard bookkeeping (chunks of other ranks' shards, straddling chunks), the
settle-loop broadcast, the MAX all-reduce of the timings, the multi_gpu gather, the merged-batch check against the CPU oracle
over the WHOLE corpus, the all-queries exact check through the redo protocol, serial and pipelined steps - with rank r on device
r % device_count and the packed records travelling over gloo (RCCL refuses two ranks on one device). The numbers of such a run
mean nothing; its control flow and its parity fields are what is tested."""
import json
import os
import socket
import subprocess
import sys

import pytest

from _util import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, [json.loads(l) for l in lines]


@pytest.mark.parametrize("world", [2, 8])
def test_bench_rank_program_runs_with_several_ranks_on_one_gpu(world):
    p, lines = _run(world, ["--dist-backend", "gloo", "--workload", "mid", "--steps", "6", "--warmup", "2", "--settle-ms", "30",
                            "--pipelined-steps", "9", "--check-queries", "16", "--dist-timeout", "240"])
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert len(lines) == 1, lines                      # rank 0 prints ONE line
    d = lines[0]
    assert "error" not in d and d["n_gpus"] == world and d["value"] > 0 and d["steps"] == 6
    mg = d["multi_gpu"]
    assert mg["rccl_ranks_seen"] == world and len(mg["per_rank_shard_rows"]) == world
    assert sum(mg["per_rank_shard_rows"]) == d["config"]["rows"] == 300_000
    assert mg["redo_batches"] == 0
    assert d["recall_at_10"] == d["ordered_match_rate"] == d["score_bit_exact_rate"] == 1.0
    assert d["check"]["anchor"] == "cpu-oracle" and d["check"]["queries"] == 16
    ex = d["exact_check"]
    assert ex["queries"] == d["config"]["batch"] and ex["redo_batches_of_the_check"] >= 1   # every record provisional -> redo on all ranks
    assert ex["recall_at_10"] == ex["ordered_match_rate"] == ex["score_bit_exact_rate"] == 1.0
    ph = d["pipelined_host"]
    assert ph["same_results_as_serial_step"] is True and ph["batches"] == 9 and ph["batches_in_flight"] == 3
    assert d["pipeline"]["fallback_queries"] == 0
    _check_diagnostics(mg, world, dry_run=True)


def _check_diagnostics(mg, world, dry_run):
    """Round 6: what makes the first real multi-GPU run readable in one shot - both query-exchange forms timed in the line, the
    per-rank phase breakdown from HIP events, the all-gather micro-latency, the ranks' device identities."""
    qx = mg["query_exchange"]
    assert qx["replicated_ms"] > 0 and qx["sharded_ms"] > 0 and qx["host_batch_steps_use"] in ("replicated", "sharded")
    if qx["selection"].startswith("auto"):
        assert qx["host_batch_steps_use"] == ("sharded" if qx["sharded_ms"] < 0.97 * qx["replicated_ms"] else "replicated")
    ph = mg["per_rank_phases_us"]
    for name in ("prep", "sample_tau", "emitting", "final_publish", "pack_and_gaps", "query_exchange", "all_gather", "merge",
                 "search_device_total"):
        assert len(ph[name]["per_rank"]) == world and ph[name]["max"] >= ph[name]["min"] >= 0.0, name
    for name in ("prep", "emitting", "final_publish", "search_device_total"):
        assert ph[name]["min"] > 0.0, name                      # every rank ran the MFMA pipeline with the phase events on
    if not dry_run:                                             # (the dry run's exchange goes through the host: no stream events)
        assert ph["all_gather"]["min"] > 0.0 and ph["merge"]["min"] > 0.0
    ag = mg["all_gather_latency_us"]
    assert ag["packed_records"] > 0 and ag["query_slices"] > 0 and ag["packed_records_bytes_per_rank"] > 0
    ids = mg["per_rank_device_identity"]
    assert [i["rank"] for i in ids] == list(range(world)) and all(i["visible_devices"] >= 1 for i in ids)
    assert mg["ranks_on_distinct_devices"] is (world == 1 or not dry_run)     # the dry run shares one GPU - and the line says so


def test_bench_one_rank_nccl_pipelined_host_step():
    """The N > 1 step with ONE rank over RCCL (--force-dist): serial value and the join-free pipelined_host leg in one line,
    both with parity."""
    p, lines = _run(1, ["--force-dist", "--workload", "c2shard8", "--steps", "20", "--warmup", "3", "--settle-ms", "50",
                        "--pipelined-steps", "30", "--check-queries", "16"])
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = lines[-1]
    assert "error" not in d and d["multi_gpu"]["rccl_ranks_seen"] == 1 and d["multi_gpu"]["backend"].startswith("nccl")
    assert d["recall_at_10"] == d["ordered_match_rate"] == d["score_bit_exact_rate"] == 1.0
    assert d["pipelined_host"]["same_results_as_serial_step"] is True
    # the timed steps' batch is resident in HBM (the default); the host-batch form is timed beside it, same results
    assert d["config"]["queries_start_in"] == "hbm" and d["pcie_inclusive_serial"]["same_results_as_value_step"] is True
    assert d["pcie_inclusive_serial"]["ms_per_step"] > 0
    ex = d["exact_check"]
    assert ex["recall_at_10"] == ex["ordered_match_rate"] == ex["score_bit_exact_rate"] == 1.0
    _check_diagnostics(d["multi_gpu"], 1, dry_run=False)


def test_bench_one_rank_nccl_sharded_query_exchange():
    """--query-exchange sharded over RCCL with one rank: the slice copy, the all-gather of the f32 slices and the shard search
    reading the gathered batch from HBM are what the timed steps run; parity fields as in the replicated form."""
    p, lines = _run(1, ["--force-dist", "--workload", "c2shard8", "--steps", "20", "--warmup", "3", "--settle-ms", "50",
                        "--pipelined-steps", "30", "--check-queries", "16", "--query-exchange", "sharded", "--queries", "host"])
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    d = lines[-1]
    assert "error" not in d and d["multi_gpu"]["query_exchange"]["timed_steps_use"] == "sharded"
    assert d["config"]["queries_start_in"] == "pinned host memory" and d["hbm_resident_serial"]["same_results_as_value_step"] is True
    assert d["multi_gpu"]["per_rank_phases_us"]["query_exchange"]["min"] > 0.0
    assert d["recall_at_10"] == d["ordered_match_rate"] == d["score_bit_exact_rate"] == 1.0
    assert d["pipelined_host"]["same_results_as_serial_step"] is True
    ex = d["exact_check"]
    assert ex["recall_at_10"] == ex["ordered_match_rate"] == ex["score_bit_exact_rate"] == 1.0


def test_bench_hung_rank_ends_with_an_error_line():
    """A rank that never joins: the others must not hang the launcher. Rank 1 is made to exit before the process group forms
    (BENCH_TEST_DROP_RANK); rank 0's init times out or its watchdog fires - either way the run ends with a non-zero status
    within the limit, and when rank 0 got as far as a JSON line it carries `error`."""
    env_extra = {"BENCH_TEST_DROP_RANK": "1"}
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--workload", "small", "--dist-timeout", "20"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert p.returncode != 0
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert all("error" in l for l in lines)
    for l in lines:          # the line of a hung run names the stage every rank reached (files: a hung rank cannot be asked)
        if l.get("per_rank_last_stage"):
            assert [r["rank"] for r in l["per_rank_last_stage"]] == [0, 1]
