"""The bench line's contract.  ``bench.make_line`` -- the code that turns what the timed region measured into the JSON line
-- is called with made-up measurements (no GPU), so a regression in the accounting fails here, not on the driver's box; the
committed evidence (profiles/r*_bench_170M_b64.json = the default ``python bench.py`` on an MI355X) is checked with the same
assertions."""
import argparse
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _made_up(batch=64, rows=170_000_000, fused=32, steps=20):
    import bench
    args = argparse.Namespace(batch=batch, top_k=10, max_answer_length=10, steps=steps, warmup=5, dist="iid", tune=[])
    n_pass = max(1, -(-2 * batch // 256)) if 2 * batch > 128 else 1
    per_launch_ms = 20.0 if batch == 64 else 30.0
    elapsed = steps * n_pass * (per_launch_ms + 0.9 + 0.6) / 1e3           # full scan + ladder scans + latency-bound rest
    line, kernel, alg_launch = bench.make_line(
        args, 1, False, rows, rows, elapsed, scan_ms=steps * n_pass * per_launch_ms, scan_launches=steps * n_pass,
        ladder_ms=steps * n_pass * 0.9, ladder_launches=steps * n_pass * 3,
        stats={"fused_stride": fused, "certified_fast": 2 * batch, "rows": 2 * batch}, pairs=1000, triggers=500, n_uncert=0)
    return line, kernel, alg_launch


def _check_consistent(d, need_cpu_baseline=True):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline") + (("cpu_baseline",) if need_cpu_baseline else ()):
        assert key in d, key
    assert d["metric"] == "queries/sec" and d["unit"] == "queries/sec" and d["higher_is_better"] is True
    assert d["data"] == "synthetic" and d["dtype"] == "int8" and d["vs_baseline"] is None
    assert "model" not in d["config"]
    B = d["config"]["batch"]
    assert abs(d["value"] - B / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6               # value = queries / wall time
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] / 1e3) / 1e9) / r["achieved"] < 1e-6
    assert r["avg_launch_ms"] <= d["ms_per_step"]                                          # the kernel fits inside the step
    assert r["achieved"] <= r["peak"]
    # algorithmic bytes per LAUNCH: the rows the launch reads (all of them, or all but every S-th tile when the finest ladder
    # level is fused into the scan) x 768 B + query rows + results
    rows = r.get("rows_read_by_the_launch", d["config"]["rows_per_gpu"])
    assert rows <= d["config"]["rows_per_gpu"]
    q = 2 * B / d["config"]["scan_launches_per_step"]
    assert r["algorithmic_bytes_per_launch"] == rows * 768 + q * 768 * 4 + q * d["config"]["top_k"] * 12
    if r["traffic"] is not None:                                                           # measured HBM bytes per launch
        assert 0.98 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10
    assert d["uncertified_rows_all_timed_steps"] == 0
    return r


@pytest.mark.parametrize("batch", [64, 128, 256, 512])
def test_make_line_accounting(batch):
    d, kernel, alg_launch = _made_up(batch)
    r = _check_consistent(d, need_cpu_baseline=False)
    assert d["n_gpus"] == 1 and ("configs[1]" in d["config"]["workload"] or batch != 64)
    assert kernel == f"dph_scan_kernel<{1 if batch == 64 else 2}, 4, false, 0, 1, false>" and alg_launch == r["algorithmic_bytes_per_launch"]
    pb = r["per_batch"]
    # per batch: the WHOLE dump once (not the launch's 31/32) + queries + results ...
    assert pb["algorithmic_bytes"] >= d["config"]["rows_total"] * 768
    assert pb["algorithmic_bytes"] == d["config"]["rows_per_gpu"] * 768 + 2 * batch * 768 * 4 + 2 * batch * 10 * 12
    # ... over ALL scan launches of the batch: full scans (ROLE 0) + ladder levels (ROLE 1)
    assert abs(pb["scan_ms"] - (pb["full_scan_ms"] + pb["ladder_scan_ms"])) < 1e-9 and pb["ladder_scan_ms"] > 0
    assert abs(pb["achieved"] - pb["algorithmic_bytes"] / (pb["scan_ms"] / 1e3) / 1e9) / pb["achieved"] < 1e-9
    assert pb["step_frac"] < pb["frac"] <= 1.0
    assert abs(d["fixed_ms_per_step"] - (d["ms_per_step"] - pb["full_scan_ms"])) < 1e-9
    m = r["mfma_int8"]
    assert abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-12 and m["peak"] == 5000.0


def test_make_line_without_a_fused_level_prices_the_whole_dump_per_launch():
    d, _, _ = _made_up(64, fused=0)
    assert d["roofline"]["rows_read_by_the_launch"] == d["config"]["rows_per_gpu"]


def _evidence():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_170M_b64.json")))
    assert files
    with open(files[-1]) as f:                       # the latest round's default run
        return json.loads(f.read().strip().splitlines()[-1]), os.path.basename(files[-1])


def test_committed_bench_line_keeps_the_contract():
    d, name = _evidence()
    _check_consistent(d)
    assert d["n_gpus"] == 1 and "configs[1]" in d["config"]["workload"]
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert "queries/sec" in json.dumps(base.get("metric", base))
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    for k in ("recall_at_1", "recall_at_5", "recall_at_10"):
        assert d[k] == 1.0
    if not name.startswith("r03"):                   # from round 4 on: the whole dump per batch, every row checked
        assert d["roofline"]["per_batch"]["algorithmic_bytes"] >= d["config"]["rows_total"] * 768
        assert d["recall_rows_checked"] == 2 * d["config"]["batch"]


def test_default_step_count_is_the_reference_benchmark_set_in_eval_batches():
    """the timed query count follows the reference's own benchmark set -- scripts/benchmark/data/
    nq_1000_dev_denspi.json, 1000 NQ-dev questions -- cut into eval batches of 64 (options.py eval_batch_size): 15 full batches, and
    run_demo.py:329-352 excludes the first 5 batches from its timing.  (The file itself only exists next to the reference.)"""
    import sys
    import bench
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.batch == 64 and a.warmup == 5
    path = "/root/reference/scripts/benchmark/data/nq_1000_dev_denspi.json"
    n_questions = len(json.load(open(path))["data"]) if os.path.exists(path) else 1000
    assert n_questions == 1000 and a.steps == n_questions // a.batch
