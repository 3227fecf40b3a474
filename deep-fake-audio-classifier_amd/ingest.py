"""Feature ingest sized for the GPU.

`features.pkl` is a pandas pickle of per-row torch tensors: un-pickling it and indexing it row by row tops out around
10^3 utterances/s, two orders of magnitude below what one MI355X consumes.  `convert()` rewrites it ONCE into a flat
`[N, 180, 321]` array (.npy, fp32 or bf16-as-uint16) plus an uttid / label index (.json); `FlatFeatures` memory-maps
that file, so a loader touches only the bytes it ships and `FlatBatcher` can stream pinned slices to the device.
The pickle reader (dataset.py) stays for drop-in use."""
from __future__ import annotations

import json
import os

import numpy as np
import pandas as pd
import torch


def convert(features_path: str, out_prefix: str, labels_path: str | None = None, dtype: str = "fp32") -> dict:
    """features.pkl (+ labels.pkl) -> <out_prefix>.npy + <out_prefix>.json.  Rows keep the merge order of the
    reference's AudioDeepfakeDataset (inner merge on uttid, src/dataset.py:28)."""
    if dtype not in ("fp32", "bf16"):
        raise ValueError("dtype must be 'fp32' or 'bf16'")
    table = pd.read_pickle(features_path)
    if "uttid" not in table.columns:
        raise ValueError("features.pkl must contain 'uttid'")
    labels = None
    if labels_path is not None:
        table = pd.merge(table, pd.read_pickle(labels_path), on="uttid", how="inner").reset_index(drop=True)
        labels = [int(v) for v in table["label"].values]
    n = len(table)
    shape = tuple(table["features"].iloc[0].shape) if n else (180, 321)
    np_dtype = np.float32 if dtype == "fp32" else np.uint16
    arr = np.lib.format.open_memmap(out_prefix + ".npy", mode="w+", dtype=np_dtype, shape=(n, *shape))
    for i, f in enumerate(table["features"]):
        t = f.float()
        if tuple(t.shape) != shape:
            raise ValueError(f"row {i}: feature shape {tuple(t.shape)} differs from {shape} (fixed-size features expected)")
        arr[i] = t.numpy() if dtype == "fp32" else t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    arr.flush()
    meta = {"uttid": [str(u) for u in table["uttid"].values], "labels": labels, "dtype": dtype, "shape": [n, *shape]}
    with open(out_prefix + ".json", "w") as fh:
        json.dump(meta, fh)
    return meta


class FlatFeatures:
    """Memory-mapped view of a converted feature file.  `.tensor()` is a zero-copy torch view of the whole array
    (pages are read on demand), `.labels` a float tensor or None, `.uttids` the id list."""

    def __init__(self, prefix: str):
        with open(prefix + ".json") as fh:
            self.meta = json.load(fh)
        self.array = np.load(prefix + ".npy", mmap_mode="r")
        self.uttids = self.meta["uttid"]
        self.labels = None if self.meta["labels"] is None else torch.tensor(self.meta["labels"], dtype=torch.float32)

    def __len__(self):
        return self.array.shape[0]

    def tensor(self) -> torch.Tensor:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # read-only memmap -> torch warns about non-writable storage
            t = torch.from_numpy(self.array)
        return t.view(torch.bfloat16) if self.meta["dtype"] == "bf16" else t


def main(argv=None):
    import argparse
    p = argparse.ArgumentParser(description="Convert features.pkl into a flat memory-mappable array.")
    p.add_argument("--features", required=True)
    p.add_argument("--labels", default=None)
    p.add_argument("--out-prefix", required=True)
    p.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    a = p.parse_args(argv)
    meta = convert(a.features, a.out_prefix, a.labels, a.dtype)
    print(f"wrote {a.out_prefix}.npy {meta['shape']} {meta['dtype']} ({os.path.getsize(a.out_prefix + '.npy') >> 20} MiB)")


if __name__ == "__main__":
    main()
