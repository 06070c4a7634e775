"""Run with an interpreter that has h5py (here: /opt/conda/bin/python3.9): writes the toy dump in the REFERENCE's
on-disk layout (phrase/0-1.hdf5 + start/<index>/idx2id.hdf5), following embed_utils.py:235-246 and
build_phrase_index.py:268-276."""
import sys

import h5py
import numpy as np

npz, out_dir = sys.argv[1], sys.argv[2]
z = np.load(npz)
import os
os.makedirs(os.path.join(out_dir, "phrase"), exist_ok=True)
os.makedirs(os.path.join(out_dir, "start", "toy_flat_none"), exist_ok=True)
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
with h5py.File(os.path.join(out_dir, "start", "toy_flat_none", "idx2id.hdf5"), "w") as f:
    g = f.create_group("0")
    g.create_dataset("doc", data=doc)
    g.create_dataset("word", data=word)
    g.attrs["offset"] = 0
