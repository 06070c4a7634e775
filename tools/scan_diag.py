#!/usr/bin/env python3
"""Where does the time of the 256-row filter scan go?  Builds copies of libdph whose scan kernel leaves one ingredient of the
streaming loop out (-DDPH_SCAN_DIAG=bits, dph_scan.hip) and times dph_debug_scan_time with each: the difference to the
product kernel is what that ingredient costs.  The variant libraries compute garbage; nothing but this tool loads them.

  build (no GPU needed):  python tools/scan_diag.py --build
  run on the GPU box:     python tools/scan_diag.py --rows 170000000 --out gpurun_out/scan_diag.json
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {0: "product", 1: "no barrier", 2: "no vmcnt wait", 3: "no barrier, no vmcnt wait", 4 + 8: "no feed (no loads, no staging writes)",
            4 + 8 + 1 + 2: "no feed, no barrier", 4 + 8 + 1 + 2 + 32: "MFMA + fragment reads only", 4 + 8 + 1 + 2 + 16 + 32: "MFMA only",
            32: "no threshold max", 8: "no staging writes", 4: "no global loads", 64: "staging writes from VGPRs", 128: "staging writes as 2 x b64",
            64 + 4: "no global loads, staging writes from VGPRs", 256: "tile loads without the nt hint",
            512: "s_setprio 3 / 0 around every MFMA", 1024: "matrix work as 2 x v_mfma_i32_16x16x64_i8 per k-step and group (results discarded)",
            1024 + 4 + 8 + 1 + 2 + 16 + 32: "16x16x64 MFMA only (no feed, no fragment reads, no threshold)"}
OUT_DIR = os.path.join(ROOT, "tools", "ubench")


def lib_path(bits):
    return os.path.join(OUT_DIR, f"libdph_diag{bits}.so")


def build(only=None):
    from densephrases_amd.build import CSRC, EXTRA_FLAGS, FLAGS, SOURCES, build as build_product
    build_product(verbose=False)
    for bits in VARIANTS:
        if bits == 0 or (only and bits not in only):
            continue
        obj = os.path.join(OUT_DIR, f"dph_scan_diag{bits}.o")
        # (variants 512 / 1024 are experiments ON the 32 x 32 x 32 kernels of rounds 1-5: built with -DDPH_SCAN_X16=0; their results are
        # garbage like every variant's -- the product's query fragments are in the 16 x 16 x 64 order)
        x16 = ["-DDPH_SCAN_X16=0"] if bits & (512 | 1024) else []
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA_FLAGS["dph_scan.hip"] + x16 + [f"-DDPH_SCAN_DIAG={bits}", "-c",
                        os.path.join(CSRC, "dph_scan.hip"), "-o", obj], check=True, cwd=CSRC)
        objs = [obj if s == "dph_scan.hip" else os.path.join(CSRC, s[:-4] + ".o") for s in SOURCES]
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(bits)] + objs, check=True)
        os.remove(obj)
        print("built", lib_path(bits), file=sys.stderr)


def run_one(rows, n_q, iters, scheds=(0,)):
    import numpy as np
    from densephrases_amd import Shard
    s = Shard(rows, device=0)
    s.fill_synthetic(seed=42, kind=0)
    s.finalize()
    x = np.random.default_rng(0).normal(0, 0.5, (n_q, 768)).astype(np.float32)
    out = {}
    for sch in scheds:
        s.set_tuning("scan_sched", sch)
        out[str(sch)] = [float(v) for v in s.debug_scan_time(x, iters)]
    s.close()
    print(json.dumps({"ms": out[str(scheds[0])], "by_sched": out}))


def run_scheds(rows, iters, out_path):
    """the hand-over schedules of the product kernel (tuning key scan_sched) at 128 and 256 query rows, interleaved A/B/A"""
    import numpy as np
    from densephrases_amd import Shard
    s = Shard(rows, device=0)
    s.fill_synthetic(seed=42, kind=0)
    s.finalize()
    res = []
    for n_q in (256, 128):
        x = np.random.default_rng(0).normal(0, 0.5, (n_q, 768)).astype(np.float32)
        for rep in range(2):
            for sch in (0, 1, 2):
                s.set_tuning("scan_sched", sch)
                ms = [float(v) for v in s.debug_scan_time(x, iters)]
                med = sorted(ms[1:])[len(ms[1:]) // 2]
                ops = 2.0 * (256 if n_q > 128 else 128) * 768 * rows
                res.append({"n_q": n_q, "sched": sch, "rep": rep, "ms": ms, "median_ms_after_first": med,
                            "int8_frac_of_5000": ops / med / 1e9 / 5000.0, "hbm_tb_s": rows * 768 / med / 1e9})
                print(json.dumps(res[-1]), file=sys.stderr, flush=True)
    s.close()
    out = {"rows": rows, "schedules": res}
    if out_path:
        with open(out_path, "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--rows", type=int, default=170_000_000)
    ap.add_argument("--n_q", type=int, default=256)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--only", type=int, nargs="*")
    ap.add_argument("--out", default="")
    ap.add_argument("--one", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--scheds", action="store_true", help="time the hand-over schedules 0 / 1 / 2 of the product kernel instead of the variants")
    a = ap.parse_args()
    if a.build:
        return build(a.only)
    if a.scheds:
        return run_scheds(a.rows, a.iters, a.out)
    if a.one:
        return run_one(a.rows, a.n_q, a.iters)
    res = []
    for bits, name in VARIANTS.items():
        if a.only and bits not in a.only:
            continue
        env = dict(os.environ)
        if bits:
            if not os.path.exists(lib_path(bits)):
                continue
            env["DPH_LIBRARY"] = lib_path(bits)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", "--rows", str(a.rows), "--n_q", str(a.n_q), "--iters", str(a.iters)],
                           env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            res.append({"bits": bits, "variant": name, "error": r.stderr[-300:]})
            continue
        ms = json.loads(r.stdout.strip().splitlines()[-1])["ms"]
        steady = sorted(ms[1:])[len(ms[1:]) // 2]
        ops = 2.0 * (256 if a.n_q > 128 else 128) * 768 * a.rows
        res.append({"bits": bits, "variant": name, "ms": ms, "median_ms_after_first": steady, "int8_top_s": ops / steady / 1e9,
                    "frac_of_5000": ops / steady / 1e9 / 5000.0, "hbm_tb_s": a.rows * 768 / steady / 1e9})
        print(json.dumps(res[-1]), file=sys.stderr, flush=True)
    out = {"rows": a.rows, "n_q": a.n_q, "variants": res}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
