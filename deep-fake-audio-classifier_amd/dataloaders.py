"""Loaders -- counterpart of src/dataloaders.py:8-59 (same function names, arguments and defaults).

`make_loader` / `create_dataloaders` return ordinary torch DataLoaders over `AudioDeepfakeDataset`, so reference
scripts keep working.  `FlatBatcher` is the ingest path sized for the GPU: one contiguous (optionally pinned) tensor,
batches are slices, host->device copies are asynchronous on a side stream and double-buffered.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import dataset as ds


def create_dataloaders(train_features_path, train_labels_path, dev_features_path, dev_labels_path,
                       test_features_path, batch_size=32, num_workers=2):
    """(train_loader [shuffled], dev_loader, test_loader [features only]) -- src/dataloaders.py:8-52."""
    train = ds.AudioDeepfakeDataset(train_features_path, train_labels_path)
    dev = ds.AudioDeepfakeDataset(dev_features_path, dev_labels_path)
    test = ds.AudioDeepfakeDataset(test_features_path, None)
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=True)
    return (DataLoader(train, shuffle=True, **kw), DataLoader(dev, shuffle=False, **kw),
            DataLoader(test, shuffle=False, **kw))


def make_loader(features_path, labels_path, batch_size=32, num_workers=2, shuffle=False):
    """src/dataloaders.py:55-59."""
    return DataLoader(ds.AudioDeepfakeDataset(features_path, labels_path), batch_size=batch_size, shuffle=shuffle,
                      num_workers=num_workers)


def train_shard_indices(perm: torch.Tensor, batch_size: int, rank: int = 0, world: int = 1) -> torch.Tensor:
    """Sample indices of `rank` for one data-parallel training epoch over the permutation `perm`.

    Global batch g is perm[g*world*bs : (g+1)*world*bs]; rank r takes rows [r*bs, (r+1)*bs) of it.  With world > 1 the
    permutation is padded by wrap-around to a multiple of world*bs (DistributedSampler semantics), so EVERY rank runs
    the same number of steps with the same local batch size: one all-reduce per step on every rank (no hang on a short
    last shard) and the 1/world gradient scale is exact.  world == 1 keeps the reference's ragged last batch
    (src/train.py:61: the DataLoader's final short batch)."""
    n = perm.numel()
    if world <= 1 or n == 0:
        return perm
    gb = world * batch_size
    total = -(-n // gb) * gb
    reps = -(-total // n)
    ext = perm.repeat(reps)[:total] if reps > 1 else perm
    return ext.view(-1, world, batch_size)[:, rank].reshape(-1)


class FlatBatcher:
    """Iterate (features [b,180,321] on `device`, labels [b] on `device` or None) over a stacked dataset.

    rank/world shard the utterance range contiguously (rank r gets [r*ceil(N/world), ...)): the data-parallel
    inference partition of SURVEY.md section 8(e); order inside a shard is preserved so host code can concatenate.
    """

    def __init__(self, features: torch.Tensor, labels: torch.Tensor | None, batch_size: int, device="cuda",
                 rank: int = 0, world: int = 1, dtype: torch.dtype | None = None):
        n = features.shape[0]
        per = -(-n // world)
        self.lo, self.hi = min(rank * per, n), min((rank + 1) * per, n)
        self.features, self.labels = features, labels
        self.batch_size, self.device, self.dtype = batch_size, torch.device(device), dtype
        self._registered = None

    def __len__(self):
        return -(-(self.hi - self.lo) // self.batch_size)

    # A pageable source (the memory-mapped flat file) makes every H2D copy block the host for its duration, so the next forward
    # is launched only after the next batch's copy has finished (measured 0.85 ms per 256-utterance batch for 0.67 ms of kernels).
    # Registering this rank's share of the mapping with the HIP runtime (hipHostRegister: the pages are pinned where they are, no
    # staging copy) makes the copies asynchronous; above `max_register_bytes` (or if the runtime refuses) the pageable path stays.
    max_register_bytes = 16 << 30

    def _register(self):
        f = self.features
        if self._registered is not None or self.device.type != "cuda" or f.device.type != "cpu" or f.is_pinned() or self.hi <= self.lo:
            return
        try:
            if not f[self.lo:self.hi].is_contiguous():
                return
            ptr = f[self.lo:self.hi].data_ptr()
            nbytes = (self.hi - self.lo) * f[0].numel() * f.element_size()
            if nbytes > self.max_register_bytes:
                return
            if int(torch.cuda.cudart().cudaHostRegister(ptr, nbytes, 0)) == 0:
                self._registered = ptr
        except Exception:   # noqa: BLE001 -- an optimisation only
            self._registered = None

    def _unregister(self):
        if self._registered is not None:
            try:
                torch.cuda.synchronize(self.device)
                torch.cuda.cudart().cudaHostUnregister(self._registered)
            except Exception:   # noqa: BLE001
                pass
            self._registered = None

    def __del__(self):
        self._unregister()

    def _put(self, lo, hi, stream):
        with torch.cuda.stream(stream):
            f = self.features[lo:hi].to(self.device, non_blocking=True)
            if self.dtype is not None:
                f = f.to(self.dtype)
            l = None if self.labels is None else self.labels[lo:hi].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        return f, l, ev

    def __iter__(self):
        if self.device.type != "cuda":
            for lo in range(self.lo, self.hi, self.batch_size):
                hi = min(lo + self.batch_size, self.hi)
                yield self.features[lo:hi], (None if self.labels is None else self.labels[lo:hi])
            return
        self._register()
        copy_stream = torch.cuda.Stream(self.device)
        starts = list(range(self.lo, self.hi, self.batch_size))
        nxt = self._put(starts[0], min(starts[0] + self.batch_size, self.hi), copy_stream) if starts else None
        try:
            for i, lo in enumerate(starts):
                f, l, ev = nxt
                if i + 1 < len(starts):
                    nlo = starts[i + 1]
                    nxt = self._put(nlo, min(nlo + self.batch_size, self.hi), copy_stream)
                torch.cuda.current_stream(self.device).wait_event(ev)
                f.record_stream(torch.cuda.current_stream(self.device))
                if l is not None:
                    l.record_stream(torch.cuda.current_stream(self.device))
                yield f, l
        finally:
            self._unregister()


class IndexedFlatBatcher:
    """Batches of ARBITRARY rows of a flat [N,180,321] source -- the data-parallel TRAINING loader (the
    reference's shuffled DataLoader, src/dataloaders.py:55-59 / src/train.py:279-285, on every rank's share).

    `features` is any row-indexable tensor: the zero-copy view of a memory-mapped flat file (ingest.FlatFeatures.tensor(): a
    rank then reads ONLY the rows it trains on -- 1/world of the set per epoch, never the whole pickle) or an in-memory stack.
    `indices` are this rank's sample indices for the epoch in consumption order (dataloaders.train_shard_indices).  Each batch is
    gathered on the host into a (pinned) staging buffer and copied to the device on a side stream, one batch ahead of the
    consumer.  `rows_fetched` / `bytes_fetched` count what this rank pulled from the source."""

    def __init__(self, features: torch.Tensor, labels: torch.Tensor | None, indices: torch.Tensor, batch_size: int,
                 device="cuda", dtype: torch.dtype | None = None, pin: bool = True):
        self.features, self.labels = features, labels
        self.indices = indices.to(torch.int64).reshape(-1)
        self.batch_size, self.device, self.dtype = int(batch_size), torch.device(device), dtype
        self.pin = bool(pin) and self.device.type == "cuda"
        self.rows_fetched = 0
        self.bytes_fetched = 0
        self._row_bytes = int(features[0].numel() * features.element_size()) if len(features) else 0
        self._stage, self._stage_ev = [None, None], [None, None]      # two persistent (pinned) staging buffers + their last H2D events
        self._pool = None

    copy_threads = 4          # worker threads of the row gather (plain memcpys; 1 = on the calling thread)

    def __len__(self):
        return -(-self.indices.numel() // self.batch_size)

    def _gather(self, lo, hi, slot=None):
        """Rows indices[lo:hi] of the source.  slot = None: a fresh tensor (CPU consumers); slot = k: copied row by row into the
        k-th of two persistent pinned staging buffers -- `features[idx]` allocates (and page-faults) a new 59 MB tensor per fp32
        batch and `pin_memory()` copies it again: 50 ms per 256 utterances on the GPU box, ten training steps' worth; 256 plain
        231 KB row copies into a buffer that already exists take 2.8 ms (tools/gpu_loader_probe2.py)."""
        idx = self.indices[lo:hi]
        n = int(idx.numel())
        self.rows_fetched += n
        self.bytes_fetched += n * self._row_bytes
        lab = None if self.labels is None else self.labels[idx]
        if slot is None:
            return self.features[idx], lab              # fancy index: copies exactly these rows out of the (memory-mapped) source
        if self._stage[slot] is None:
            buf = torch.empty((self.batch_size, *self.features.shape[1:]), dtype=self.features.dtype)
            self._stage[slot] = buf.pin_memory() if self.pin else buf
        if self._stage_ev[slot] is not None:
            self._stage_ev[slot].synchronize()           # the H2D copy that last read this buffer has finished (long ago)
        rows = self._stage[slot][:n]
        # one plain memcpy per row on this thread (ctypes.memmove): torch's copy / index kernels fan a 231 KB row out over every
        # CPU the machine REPORTS (128 on the GPU box, 16 of them ours: 112 ms per batch), numpy.take buffers (14 ms)
        import ctypes
        src, dst, rb = self.features.data_ptr(), rows.data_ptr(), self._row_bytes
        if self.features.is_contiguous():
            rl = idx.tolist()

            def copy_rows(j0, j1):                      # (ctypes releases the GIL inside memmove: the chunks run in parallel)
                for j in range(j0, j1):
                    ctypes.memmove(dst + j * rb, src + rl[j] * rb, rb)
            nw = self.copy_threads if n >= 4 * self.copy_threads else 1
            if nw > 1:
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(max_workers=nw)
                step = -(-n // nw)
                list(self._pool.map(lambda k: copy_rows(k * step, min(n, (k + 1) * step)), range(nw)))
            else:
                copy_rows(0, n)
        else:
            for j, r in enumerate(idx.tolist()):
                rows[j].copy_(self.features[r])
        return rows, lab

    def _put(self, lo, hi, stream, slot=0):
        rows, lab = self._gather(lo, hi, slot)
        with torch.cuda.stream(stream):
            f = rows.to(self.device, non_blocking=True)
            if self.dtype is not None:
                f = f.to(self.dtype)
            l = None if lab is None else lab.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        self._stage_ev[slot] = ev
        return f, l, ev, rows                           # `rows` keeps the pinned staging buffer alive until the copy is consumed

    def __iter__(self):
        n = self.indices.numel()
        starts = list(range(0, n, self.batch_size))
        if self.device.type != "cuda":
            for lo in starts:
                rows, lab = self._gather(lo, min(lo + self.batch_size, n))
                yield (rows if self.dtype is None else rows.to(self.dtype)), lab
            return
        copy_stream = torch.cuda.Stream(self.device)
        nxt = self._put(starts[0], min(starts[0] + self.batch_size, n), copy_stream, 0) if starts else None
        for i, lo in enumerate(starts):
            f, l, ev, _keep = nxt
            if i + 1 < len(starts):
                nlo = starts[i + 1]
                nxt = self._put(nlo, min(nlo + self.batch_size, n), copy_stream, (i + 1) & 1)
            torch.cuda.current_stream(self.device).wait_event(ev)
            f.record_stream(torch.cuda.current_stream(self.device))
            if l is not None:
                l.record_stream(torch.cuda.current_stream(self.device))
            yield f, l


class ResidentBatcher:
    """The training set held in HBM: the flat [N,180,321] source is uploaded ONCE (in chunks, optionally rounded to bf16 on the
    device) and a batch is a row gather on the device (`index_select`: 59 MB in ~40 us).  An MI355X has 288 GB: 100 k utterances
    are 23 GB in fp32, so for the reference's data sets the host never takes part in an epoch again -- the loop runs at the speed
    of the training step (the host-fed IndexedFlatBatcher delivers ~100 k utterances/s per rank; the CNN1D step consumes 430 k).
    Same interface and batch contents as IndexedFlatBatcher(features, labels, indices, batch_size): `epoch(indices)` returns the
    iterable for one epoch's index order (dataloaders.train_shard_indices)."""

    def __init__(self, features: torch.Tensor, labels: torch.Tensor | None, batch_size: int, device="cuda",
                 dtype: torch.dtype | None = None, chunk_rows: int = 2048):
        self.device, self.batch_size = torch.device(device), int(batch_size)
        n = features.shape[0]
        store = dtype or features.dtype
        self.features = torch.empty((n, *features.shape[1:]), dtype=store, device=self.device)
        for lo in range(0, n, chunk_rows):                   # bounded host staging: the source is a memory map
            hi = min(n, lo + chunk_rows)
            self.features[lo:hi].copy_(features[lo:hi].to(self.device, non_blocking=False))
        self.labels = None if labels is None else labels.to(self.device)
        self.bytes_resident = self.features.numel() * self.features.element_size()

    @staticmethod
    def fits(features: torch.Tensor, device, dtype: torch.dtype | None = None, fraction: float = 0.5) -> bool:
        """True when the whole set takes at most `fraction` of the device's currently free memory."""
        dev = torch.device(device)
        if dev.type != "cuda":
            return False
        es = torch.empty((), dtype=dtype or features.dtype).element_size()
        free, _total = torch.cuda.mem_get_info(dev)
        return features.numel() * es <= fraction * free

    def epoch(self, indices: torch.Tensor):
        idx = indices.to(self.device, dtype=torch.int64).reshape(-1)
        bs, feats, labels = self.batch_size, self.features, self.labels

        class _Epoch:
            def __len__(self_inner):
                return -(-idx.numel() // bs)

            def __iter__(self_inner):
                for lo in range(0, idx.numel(), bs):
                    rows = idx[lo:lo + bs]
                    yield feats.index_select(0, rows), (None if labels is None else labels.index_select(0, rows))
        return _Epoch()


def open_flat(features_path: str, labels_path: str | None, cache_dir: str, rank: int = 0, world: int = 1, tag: str = "set"):
    """(features tensor [N,180,321] zero-copy over a memory-mapped flat file, labels [N] float32 or None, uttids) for the
    data-parallel drivers.  `features_path` is either a flat prefix made by `python -m dfa_amd.ingest` (<prefix>.npy +
    <prefix>.json: nothing is un-pickled by anyone) or the reference's features.pkl: then RANK 0 ALONE converts it once into
    `cache_dir/<tag>_flat.*` (ingest.convert), the other ranks wait at a barrier and map the result -- instead of every rank
    un-pickling, stacking and pinning the whole set."""
    import os
    from . import ingest
    import torch.distributed as dist
    if os.path.exists(features_path + ".npy") and os.path.exists(features_path + ".json"):
        prefix = features_path
    else:
        prefix = os.path.join(cache_dir, f"{tag}_flat")
        if rank == 0:
            os.makedirs(cache_dir, exist_ok=True)
            ingest.convert(features_path, prefix, labels_path, dtype="fp32")
        if world > 1 and dist.is_available() and dist.is_initialized():
            dist.barrier()
    ff = ingest.FlatFeatures(prefix)
    return ff.tensor(), ff.labels, ff.uttids
