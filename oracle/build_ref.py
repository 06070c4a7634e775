#!/usr/bin/env python3
"""Recipe for oracle/_ref/: the reference's own python files of the hot path, COMPILED (CPython byte code) from the sources
where they lie under /root/reference.  Test infrastructure, not product: nothing under densephrases_amd/ reads oracle/_ref.

Why: /root/reference does not exist on the GPU box, and reference sources must not be copied into this repository.  The byte
code is a build output (like a .so compiled from reference C files would be): git-ignored, shipped to the GPU box with the
working tree, loaded there by oracle/refshim with a sourceless loader -- so the `-m gpu` tests can run the reference's
UNMODIFIED ``MIPS`` (index.py), ``DensePhrases`` (model.py), ``evaluate`` (eval_phrase_retrieval.py), ``load_phrase_index`` /
``get_query2vec`` / ``load_qa_pairs`` (open_utils.py), ``Options`` (options.py), the metric functions (eval_utils.py) and
``get_top_phrases`` / ``annotate_phrase_vecs`` (train_query.py) over libdph on a real MI355X.

    python -m oracle.build_ref          # needs /root/reference; __graft_entry__.build() calls build_ref() when it exists
"""
from __future__ import annotations

import os
import py_compile
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_ROOT = "/root/reference"
OUT_DIR = os.path.join(ROOT, "oracle", "_ref")
# module name -> path under the reference tree
REF_FILES = {
    "densephrases.index": "densephrases/index.py",
    "densephrases.model": "densephrases/model.py",
    "densephrases.options": "densephrases/options.py",
    "densephrases.utils.open_utils": "densephrases/utils/open_utils.py",
    "densephrases.utils.eval_utils": "densephrases/utils/eval_utils.py",
    "eval_phrase_retrieval": "eval_phrase_retrieval.py",
    "train_query": "train_query.py",                  # the second caller of MIPS.search: get_top_phrases, annotate_phrase_vecs
}


def bin_path(modname: str) -> str:
    return os.path.join(OUT_DIR, modname + ".refpyc")


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "densephrases"))


def build_ref(verbose: bool = False) -> int:
    """compile every file of REF_FILES that is newer than its output; returns the number of files compiled"""
    if not have_reference():
        return 0
    os.makedirs(OUT_DIR, exist_ok=True)
    n = 0
    for mod, rel in REF_FILES.items():
        src, dst = os.path.join(REFERENCE_ROOT, rel), bin_path(mod)
        if os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src):
            continue
        # dfile: the path tracebacks show (the source is not on the box that runs the byte code)
        py_compile.compile(src, cfile=dst, dfile=f"<reference>/{rel}", doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        n += 1
        if verbose:
            print(f"compiled {src} -> {dst}", file=sys.stderr)
    with open(os.path.join(OUT_DIR, "PYTHON_VERSION"), "w") as f:
        f.write(".".join(map(str, sys.version_info[:3])) + "\n")
    return n


if __name__ == "__main__":
    if not have_reference():
        sys.exit("no /root/reference here: oracle/_ref can only be built in the build container")
    print("compiled", build_ref(verbose=True), "file(s) into", OUT_DIR)
