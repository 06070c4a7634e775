#!/usr/bin/env python3
"""A / B on one box: the full filter scan (dph_debug_scan_time: every tile streamed and multiplied, nothing emitted) with the int8 matrix
work as v_mfma_i32_16x16x64_i8 (the product, round 6) against the 32 x 32 x 32 form of rounds 1-5 (a copy of the library built with
-DDPH_SCAN_X16=0: tools/ubench/libdph_diag_x16off.so), 128 and 256 query rows, interleaved A/B/A/B in separate processes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OFF = os.path.join(ROOT, "tools", "ubench", "libdph_diag_x16off.so")


def main():
    res = []
    for rep in range(2):
        for name, lib in (("16x16x64", None), ("32x32x32", OFF)):
            for n_q in (256, 128):
                env = dict(os.environ)
                if lib:
                    env["DPH_LIBRARY"] = lib
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_diag.py"), "--one", "--rows", "170000000", "--n_q", str(n_q), "--iters", "6"],
                                   env=env, capture_output=True, text=True, timeout=600)
                if r.returncode != 0:
                    res.append({"mfma": name, "n_q": n_q, "rep": rep, "error": r.stderr[-300:]})
                    continue
                ms = json.loads(r.stdout.strip().splitlines()[-1])["ms"]
                med = sorted(ms[1:])[len(ms[1:]) // 2]
                ops = 2.0 * n_q * 768 * 170000000
                res.append({"mfma": name, "n_q": n_q, "rep": rep, "ms": ms, "median_ms_after_first": med, "int8_top_s": ops / med / 1e9, "hbm_tb_s": 170000000 * 768 / med / 1e9})
                print(json.dumps(res[-1]), file=sys.stderr, flush=True)
    print(json.dumps({"rows": 170000000, "runs": res}))


if __name__ == "__main__":
    main()
