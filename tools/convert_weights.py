#!/usr/bin/env python3
"""Convert the reference's released SeisBench-format weights into the neutral
fixture format shipped in this repo.

Input  (read-only, /root/reference/Final_models/**):
    <name>.pt.v1   = torch.save(model.state_dict())   [REF model_training/tune.ipynb:143-149]
    <name>.json.v1 = {docstring, model_args, seisbench_requirement, version, default_args}
                                                      [REF model_training/tune.ipynb:90-121]
Output (volpick_amd/weights/<model>/):
    <name>.npz     = {tensor name -> float32 ndarray}  (num_batches_tracked kept as int64)
    <name>.json    = the metadata JSON, verbatim

The weight files are DATA (trained parameters, GPL-3.0 like the reference
repository, see volpick_amd/weights/LICENSE.weights); no reference source code
is copied.  Run once in the build container:

    python tools/convert_weights.py [/root/reference]

It also writes tests/golden/weights.sha256: one SHA-256 per tensor and per metadata JSON, computed FROM THE REFERENCE'S
FILES (not from the converted copies).  tests/test_weights_pinned.py recomputes them from what the package ships, so the
only hot-path data the reference pins stays pinned.
"""
import hashlib
import json
import shutil
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
SETS = {
    # (reference sub-directory, weight-set name)
    "phasenet": [("volpick/phasenet", "volpick"), ("volpick_95train_5val/phasenet", "volpick_95train")],
    "eqtransformer": [
        ("volpick/eqtransformer", "volpick"),
        ("volpick_95train_5val/eqtransformer", "volpick_95train"),
    ],
}


def tensor_digest(name, a):
    """SHA-256 over name, dtype, shape and the little-endian bytes of one tensor."""
    a = np.ascontiguousarray(a)
    h = hashlib.sha256()
    h.update(f"{name}|{a.dtype.newbyteorder('<').str}|{','.join(map(str, a.shape))}|".encode())
    h.update(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
    return h.hexdigest()


def json_digest(meta):
    """SHA-256 of the metadata in canonical form (sorted keys, no insignificant white space)."""
    return hashlib.sha256(json.dumps(meta, sort_keys=True, separators=(",", ":"), ensure_ascii=True).encode()).hexdigest()


def digest_lines(model, name, arrays, meta):
    lines = [f"{json_digest(meta)}  {model}/{name}.json"]
    lines += [f"{tensor_digest(k, arrays[k])}  {model}/{name}.npz:{k}" for k in arrays]
    return lines


def main(ref_root="/root/reference"):
    digests = ["# SHA-256 of every tensor and metadata JSON of the four released weight sets, computed by tools/convert_weights.py",
               "# from /root/reference/Final_models/**/*.{pt,json}.v1 (tensor: name|dtype|shape|little-endian bytes; JSON: canonical form)"]
    ref = Path(ref_root) / "Final_models"
    for model, entries in SETS.items():
        out_dir = REPO / "volpick_amd" / "weights" / model
        out_dir.mkdir(parents=True, exist_ok=True)
        for sub, name in entries:
            sd = torch.load(ref / sub / f"{name}.pt.v1", map_location="cpu", weights_only=True)
            arrays = {k: v.detach().cpu().numpy() for k, v in sd.items()}
            np.savez(out_dir / f"{name}.npz", **arrays)
            meta = json.loads((ref / sub / f"{name}.json.v1").read_text())
            (out_dir / f"{name}.json").write_text(json.dumps(meta, indent=4) + "\n")
            digests += digest_lines(model, name, arrays, meta)
            n = sum(a.size for a in arrays.values() if a.dtype == np.float32)
            print(f"{model}/{name}: {len(arrays)} tensors, {n} floats")
    (REPO / "tests" / "golden" / "weights.sha256").write_text("\n".join(digests) + "\n")
    lic = Path(ref_root) / "LICENSE"
    if lic.exists():
        shutil.copy(lic, REPO / "volpick_amd" / "weights" / "LICENSE.weights")


if __name__ == "__main__":
    main(*sys.argv[1:])
