#!/usr/bin/env python3
"""One tracked file per workload from which `roofline.frac` can be recomputed: per timed launch of the dominant
kernel, the HIP-event duration (libdph's profiling pairs: dph_profile_read_each) AND the dispatch duration of a `rocprofv3
--kernel-trace` pass of the SAME command on the SAME box, warm-up dispatches excluded, the box's identity in the header.

  python tools/trace_out.py --leg flat|anisotropic|ivf4096|pq [--out profiles/r06_trace_<leg>.json] [--steps 10] [--warmup 3]
  (bench.py --trace_out PATH runs the `flat` leg -- `--dist anisotropic`: that leg -- and writes PATH)

The parent runs the leg twice in child processes -- plain, then under rocprofv3 -- and holds the three series against each other:
events without the profiler, events under the profiler, the profiler's own per-dispatch durations."""
import argparse
import glob
import json
import os
import shutil
import socket
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNEL = {"flat": "dph_scan_kernel<1, 4, false, 0, 1, false>", "anisotropic": "dph_scan_kernel<1, 4, false, 0, 1, true>",
          "ivf4096": "dph_scan_units_kernel<0, false>", "pq": "dph_coarse_scan_kernel"}


def child(a):
    import torch
    import __graft_entry__ as g
    g.build()
    import bench
    from densephrases_amd import Shard
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    args = bench.parse.__wrapped__() if hasattr(bench.parse, "__wrapped__") else argparse.Namespace(
        batch=64, top_k=10, max_answer_length=10, seed=42, dist="iid", queries="planted", tune=[], no_check=False)
    n = a.rows
    alg = None
    if a.leg in ("flat", "anisotropic"):
        from densephrases_amd.dist import ShardedSearcher
        kind = 0 if a.leg == "flat" else 4
        s = Shard(n, device=0)
        s.fill_synthetic(seed=42, kind=kind)
        ids = np.arange(n, dtype=np.int64)
        s.set_idx2id((ids // 100).astype(np.int32), (ids % 100).astype(np.int32))
        nd = n // 100
        s.set_f2o(np.arange(nd + 1, dtype=np.int32), np.arange(0, (nd + 2) * 100, 100, dtype=np.int64), np.tile(np.arange(100, dtype=np.int32), nd + 1))
        del ids
        s.finalize()
        ss = ShardedSearcher(s, 64, 10, 10, device=dev)
        batches, _ = bench.make_batches(args, 64, n, kind, a.warmup + a.steps, dev)
        step = lambda i: ss.step(batches[i])      # noqa: E731
        st = lambda: s.stats()                    # noqa: E731
    elif a.leg == "ivf4096":
        from densephrases_amd.ivf import assign_lists_resident, train_centroids_resident
        from densephrases_amd.synth import synthetic_rows
        n = n // 32 * 32
        s = Shard(n, device=0)
        s.fill_synthetic(seed=42, kind=3)
        cent, _ = train_centroids_resident(s, 4096, iters=10, return_info=True)
        assign = assign_lists_resident(s, cent)
        s.make_list_major(assign.data_ptr(), cent, stream=torch.cuda.current_stream(dev).cuda_stream)
        counts = torch.bincount(assign.to(torch.int64), minlength=4096)
        del assign
        torch.cuda.empty_cache()
        s.finalize()
        R, k = 512, 10
        rng = np.random.default_rng(7)
        xs = []
        for i in range(a.warmup + a.steps):
            rows = rng.integers(0, n, R)
            base = np.concatenate([synthetic_rows(int(r), 1, seed=42, kind=3) for r in rows]).astype(np.float32) / 20.0 - 2.0
            xs.append(torch.from_numpy((base + rng.normal(0, 0.25, base.shape)).astype(np.float32)).to(dev))
        D = torch.empty((R, k), dtype=torch.float32, device=dev)
        I = torch.empty((R, k), dtype=torch.int64, device=dev)
        status = torch.empty(R, dtype=torch.int32, device=dev)
        step = lambda i: s.search_ivf_dev(xs[i].data_ptr(), R, k, 256, D.data_ptr(), I.data_ptr(), status.data_ptr())     # noqa: E731
        # bytes of the lists probed by >= 1 row of the LAST batch, padded to whole tiles (SURVEY 8d "IVF scan")
        probe = torch.topk(xs[-1] @ torch.from_numpy(cent).to(dev).T, 256, dim=1).indices
        hit = torch.zeros(4096, dtype=torch.bool, device=dev)
        hit[probe.flatten()] = True
        alg = float((((counts + 31) // 32 * 32) * hit).sum().item()) * 768
        st = lambda: {}                           # noqa: E731
    else:
        from densephrases_amd.synth import synthetic_pq_shard
        s, A, cent, sizes = synthetic_pq_shard(n, 1 << 20, 96, device=0)
        R, k = 128, 10
        rng = np.random.default_rng(3)
        xs = [torch.from_numpy(rng.normal(0, 0.5, (R, 768)).astype(np.float32)).to(dev) for _ in range(a.warmup + a.steps)]
        D = torch.empty((R, k), dtype=torch.float32, device=dev)
        I = torch.empty((R, k), dtype=torch.int64, device=dev)
        status = torch.empty(R, dtype=torch.int32, device=dev)
        step = lambda i: s.search_ivf_dev(xs[i].data_ptr(), R, k, 256, D.data_ptr(), I.data_ptr(), status.data_ptr())     # noqa: E731
        alg = float((1 << 20) * 768 * 2 + R * 768 * 2)
        st = lambda: {}                           # noqa: E731
    s.profile_enable(True)
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    warm = len(s.profile_read_each(0))
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    ev = s.profile_read_each(0)
    stats = st()
    if a.leg in ("flat", "anisotropic"):
        tiles = (n + 31) // 32
        fused = int(stats.get("fused_stride", 0) or 0)
        launch_rows = (tiles - (tiles + fused - 1) // fused) * 32 if fused >= 2 else n
        alg = float(launch_rows * 768 + 128 * 768 * 4 + 128 * 10 * 12)
    print("DPH_TRACE " + json.dumps({"leg": a.leg, "kernel": KERNEL[a.leg], "rows": n, "warmup_dispatches": warm, "timed_dispatches": len(ev),
                                     "event_ms": [float(v) for v in ev], "ms_per_step": dt * 1e3, "algorithmic_bytes_per_launch": alg}), flush=True)


def box_id():
    out = {"hostname": socket.gethostname()}
    for key, cmd in (("serial", ["rocm-smi", "--showserial"]), ("bus", ["rocm-smi", "--showbus"]), ("product", ["rocm-smi", "--showproductname"])):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=30)
            out[key] = [ln.strip() for ln in r.stdout.splitlines() if "GPU[" in ln][:4]
        except (OSError, subprocess.TimeoutExpired) as e:
            out[key] = repr(e)[:80]
    return out


def run_child(a, profiled):
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "--leg", a.leg, "--steps", str(a.steps), "--warmup", str(a.warmup), "--rows", str(a.rows)]
    tmp = None
    if profiled:
        exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
        tmp = tempfile.mkdtemp(prefix="dph_trace_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "-d", tmp, "--"] + cmd
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=1500)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DPH_TRACE ")]
    if r.returncode != 0 or not line:
        raise RuntimeError(f"trace child failed (rc {r.returncode}): {(r.stderr or r.stdout)[-400:]}")
    rec = json.loads(line[-1][len("DPH_TRACE "):])
    if profiled:
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = cur.execute("select duration from kernels where name like ? order by start", ("%" + rec["kernel"] + "%",)).fetchall()
        d = [r_[0] / 1e6 for r_ in rows]                       # ns -> ms
        rec["rocprof_dispatches_total"] = len(d)
        rec["rocprof_ms"] = d[rec["warmup_dispatches"]: rec["warmup_dispatches"] + rec["timed_dispatches"]]
        rec["rocprof_warmup_ms"] = d[: rec["warmup_dispatches"]]
        shutil.rmtree(tmp, ignore_errors=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", choices=list(KERNEL), default="flat")
    ap.add_argument("--out", default="")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    plain = run_child(a, False)
    prof = run_child(a, True)
    avg = lambda v: float(np.mean(v)) if len(v) else None     # noqa: E731
    alg = plain["algorithmic_bytes_per_launch"]
    e0, e1, rp = avg(plain["event_ms"]), avg(prof["event_ms"]), avg(prof["rocprof_ms"])
    out = {"what": "per timed launch of the dominant kernel: HIP-event durations (libdph profiling pairs) without and under rocprofv3, and rocprofv3's own "
                   "per-dispatch durations of the same run; warm-up dispatches excluded from all three",
           "box": box_id(), "leg": a.leg, "kernel": plain["kernel"], "rows": plain["rows"], "steps": a.steps, "warmup_steps": a.warmup,
           "algorithmic_bytes_per_launch": alg,
           "events_plain_ms": plain["event_ms"], "events_under_rocprof_ms": prof["event_ms"], "rocprof_dispatch_ms": prof["rocprof_ms"],
           "rocprof_warmup_dispatch_ms": prof.get("rocprof_warmup_ms"),
           "summary": {"avg_event_plain_ms": e0, "avg_event_under_rocprof_ms": e1, "avg_rocprof_dispatch_ms": rp,
                       "rocprof_over_event_same_run": rp / e1 if (rp and e1) else None, "profiler_overhead_on_events": e1 / e0 if (e0 and e1) else None,
                       "hbm_frac_from_events_plain": alg / (e0 / 1e3) / 8e12 if e0 else None, "hbm_frac_from_rocprof": alg / (rp / 1e3) / 8e12 if rp else None,
                       "ms_per_step_plain": plain["ms_per_step"], "ms_per_step_under_rocprof": prof["ms_per_step"],
                       "dispatch_counts": {"warmup": prof["warmup_dispatches"], "timed": prof["timed_dispatches"], "in_trace": prof["rocprof_dispatches_total"]}}}
    txt = json.dumps(out)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    print(json.dumps(out["summary"]))


if __name__ == "__main__":
    main()
