#!/usr/bin/env python3
"""Audit of the scan kernel's ISA (cross-compiled, no GPU needed):
  * the hand-owned range a[236-24*NSET : 255] (aux operands, probe masks, staging sets) of every dph_scan_kernel<QB, NSET, ...> / dph_scan_units_kernel<ROLE> instantiation may only
    be touched inside ;;#ASMSTART/;;#ASMEND blocks;
  * no scratch, no spills;
  * prints the instruction mix for the record.
Usage: audit_scan_isa.py [path/to/dph_scan.hip]   (exit code 1 on a violation)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(src=None, verbose=True) -> int:
    src = src or os.path.join(ROOT, "densephrases_amd", "csrc", "dph_scan.hip")
    with tempfile.TemporaryDirectory() as tmp:
        sys.path.insert(0, ROOT)
        from densephrases_amd.build import EXTRA_FLAGS, FLAGS            # the flags the library is built with
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA_FLAGS.get("dph_scan.hip", []) +
                       ["-c", src, "-o", os.path.join(tmp, "o.o"), "-save-temps=obj"], check=True, cwd=os.path.dirname(src),
                       stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, "dph_scan-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    bad = 0
    for m in re.finditer(r"^(_Z15dph_scan_kernelILi(\d)ELi(\d)E\w+|_Z21dph_scan_units_kernelILi\d\w+|_Z22dph_coarse_scan_kernel\w+|_Z28dph_coarse_scan_teams_kernel\w+):.*?s_endpgm", asm,
                         flags=re.S | re.M):
        name, body = m.group(1), m.group(0)
        owned_from = 256 - 24 * int(m.group(3) or 4) - 4 - 32     # staging sets, probe masks / queue atomic, the aux operands: a ring of 4 tiles x 2 row halves (the unit scan runs 4 staging sets)
        in_asm, hits = False, []
        for ln in body.splitlines():
            if "#ASMSTART" in ln:
                in_asm = True
            elif "#ASMEND" in ln:
                in_asm = False
            elif not in_asm:
                for a in re.findall(r"\ba\[?(\d+)(?::(\d+))?\]?", ln.split(";")[0]):
                    lo = int(a[0])
                    hi = int(a[1]) if a[1] else lo
                    if hi >= owned_from:
                        hits.append(ln.strip())
        # the streaming loop (the basic blocks that hold a tile's 24+ MFMAs) must not contain a compiler-inserted vmcnt wait:
        # every wait on the feed is counted by hand inside the asm statements; one the compiler adds (for a load it still
        # believes to be in flight) drains the staged tiles
        in_asm, blk_mfma, blk_waits, stray = False, 0, [], []
        for ln in body.splitlines() + [".LBBend:"]:
            t = ln.strip()
            if re.match(r"^\.LBB\w+:", t):
                if blk_mfma >= 8:
                    stray += blk_waits
                blk_mfma, blk_waits = 0, []
            elif "#ASMSTART" in t:
                in_asm = True
            elif "#ASMEND" in t:
                in_asm = False
            elif "v_mfma" in t:
                blk_mfma += 1
            elif not in_asm and t.startswith("s_waitcnt") and "vmcnt" in t:
                blk_waits.append(t)
        hits += [f"compiler-inserted wait inside the streaming loop: {w}" for w in stray]
        mix = {k: len(re.findall(k, body)) for k in ("v_mfma", "ds_read_b128", "ds_write_b128", "global_load_dwordx4",
                                                      "v_accvgpr", "s_barrier", "scratch_")}
        if verbose:
            print(name[:40], f"owned a[{owned_from}:255]", mix, "VIOLATIONS" if hits else "ok")
            for h in hits[:10]:
                print("   violation:", h)
        bad += len(hits) + mix["scratch_"]
    return bad


# hot kernels outside dph_scan.hip that must not touch scratch memory either (round 4: hipcc left the filter GEMM's 256 bytes of
# staging registers per thread in scratch -- behind a lambda's reference parameter, then as arrays of HIP's uint4 struct -- and
# the kernel ran 1.19 ms instead of 0.4 without a single warning)
NO_SCRATCH = {"dph_ivf.hip": ["dph_coarse_filter_gemm_kernelILb0", "dph_coarse_filter_gemm_kernelILb1", "dph_coarse_filter_gemm2_kernel", "dph_coarse_gemm_bf16x3_pipe_kernel",
                              "dph_coarse_select_kernel", "dph_coarse_bucket_kernel", "dph_scan_units"],
              "dph_pq.hip": ["pq_adc_rows_kernelILi6", "pq_adc_kernelILi6", "pq_final_kernel", "pq_transform_kernel", "pq_lut_kernel"]}


def audit_no_scratch(verbose=True) -> int:
    """private_segment_fixed_size / vgpr spills of the kernels named in NO_SCRATCH (kernel descriptors of the cross-compiled objects)."""
    sys.path.insert(0, ROOT)
    from densephrases_amd.build import EXTRA_FLAGS, FLAGS
    csrc = os.path.join(ROOT, "densephrases_amd", "csrc")
    bad = 0
    for fname, kernels in NO_SCRATCH.items():
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA_FLAGS.get(fname, []) + ["-c", os.path.join(csrc, fname), "-o", os.path.join(tmp, "o.o"),
                            "-save-temps=obj"], check=True, cwd=csrc, stderr=subprocess.DEVNULL)
            asm = open(os.path.join(tmp, fname[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", asm, flags=re.S):
            name, meta = m.group(1), m.group(2)
            if not any(k in name for k in kernels):
                continue
            priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
            spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
            vg = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
            if verbose:
                print(f"{fname}: {name[:60]} vgprs {vg} scratch {priv} B spills {spill}", "VIOLATION" if priv or spill else "ok")
            bad += 1 if (priv or spill) else 0
    return bad


if __name__ == "__main__":
    if "--no-scratch" in sys.argv:
        sys.exit(1 if audit_no_scratch() else 0)
    sys.exit(1 if audit(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else None) else 0)
