"""Run with an interpreter that has h5py (here: /opt/conda/bin/python3.9): writes the toy dump in the REFERENCE's
on-disk layout (phrase/0-1.hdf5 + start/<index>/idx2id.hdf5), following embed_utils.py:235-246 and
build_phrase_index.py:268-276."""
import sys

import h5py
import numpy as np

npz, out_dir = sys.argv[1], sys.argv[2]
# optional third argument "split:<max_idx>": write the idx2id of a MERGED index -- two sub-indexes with id offsets 0 and
# <max_idx> (scripts/parallel/add_to_index.py:42-51 adds dump k with --offset k*max_idx; build_phrase_index.py:268-276
# writes one idx2id group per offset)
# "name:<index_name>": the directory under start/ (default toy_flat_none; a name containing "PQ" selects the RAM metadata
# branch of the reference's MIPS, index.py:33,69-76)
extra = sys.argv[3:]
split = next((int(a.split(":")[1]) for a in extra if a.startswith("split:")), None)
index_name = next((a.split(":", 1)[1] for a in extra if a.startswith("name:")), "toy_flat_none")
z = np.load(npz)
import os
os.makedirs(os.path.join(out_dir, "phrase"), exist_ok=True)
os.makedirs(os.path.join(out_dir, "start", index_name), exist_ok=True)
ids = z["doc_ids"].tolist()
with h5py.File(os.path.join(out_dir, "phrase", "0-1.hdf5"), "w") as f:
    for i, d in enumerate(ids):
        g = f.create_group(str(d))
        g.attrs["context"] = str(z["contexts"][i])
        g.attrs["title"] = str(z["titles"][i])
        g.attrs["offset"] = -2.0
        g.attrs["scale"] = 20.0
        g.create_dataset("start", data=z[f"start_{d}"])
        g.create_dataset("f2o_start", data=z[f"f2o_{d}"])
        g.create_dataset("word2char_start", data=z[f"w2cs_{d}"])
        g.create_dataset("word2char_end", data=z[f"w2ce_{d}"])
order = [d for d in sorted(ids, key=str) if z[f"start_{d}"].shape[0] > 0]
doc = np.concatenate([np.full(z[f"start_{d}"].shape[0], d, np.int32) for d in order])
word = np.concatenate([np.arange(z[f"start_{d}"].shape[0], dtype=np.int32) for d in order])
with h5py.File(os.path.join(out_dir, "start", index_name, "idx2id.hdf5"), "w") as f:
    if split is None:
        parts = [(0, doc, word)]
    else:
        half = len(order) // 2
        cut = int(sum(z[f"start_{d}"].shape[0] for d in order[:half]))
        parts = [(0, doc[:cut], word[:cut]), (split, doc[cut:], word[cut:])]
    for off, d_, w_ in parts:
        g = f.create_group(str(off))
        g.create_dataset("doc", data=d_)
        g.create_dataset("word", data=w_)
        g.attrs["offset"] = off
