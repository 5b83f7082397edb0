export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/final/pytest_gpu.txt | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/final/default_bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final/driverform_bench.json
timeout 600 python bench.py --workload c1 2>/dev/null | tail -1 > gpurun_out/final/c1_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 1 --force-dist --workload c2shard8 --cpu-seconds 0 2>/dev/null | grep "^{" | tail -1 > gpurun_out/final/c2shard8_force_dist_bench.json
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29553 bench.py --gpus 1 --force-dist --cpu-seconds 0 2>/dev/null | grep "^{" | tail -1 > gpurun_out/final/c2_force_dist_bench.json
python - <<'PY'
import json
for f in ("default","driverform","c1","c2shard8_force_dist","c2_force_dist"):
    d=json.load(open(f"gpurun_out/final/{f}_bench.json")); r=d.get("roofline") or {}
    print(f, d["value"], d["ms_per_step"], d.get("median_ms_per_step"), "other", (d.get("pcie_inclusive_serial") or d.get("hbm_resident_serial") or {}).get("ms_per_step"), "frac", r.get("frac"), "launch", r.get("avg_launch_ms"), "traffic", r.get("traffic"), "ph", (d.get("pipelined_host") or {}).get("ms_per_batch"), "rec", d.get("recall_at_10"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
