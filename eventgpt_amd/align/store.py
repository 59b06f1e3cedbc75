"""Chunked store for paired hidden states + the loaders that stream it.

On-disk layout (identical to the reference extractor's, so either side can read
the other's data — pipeline/feature_extraction/extract_hidden_states.py:676-880):

    <dir>/index.json   {total_samples, chunk_size, chunks:[{path,start_idx,n_samples,
                        [source_chunk]}], config, created}
    <dir>/chunks/chunk_%06d.pt
                       {egpt_hidden [N,S,H] f32, vl_hidden [N,S,H] f32, seq_lens [N],
                        metadata [N] of dicts, chunk_id, n_samples}

Additions: chunk files are written atomically (temp + rename) so a kill mid-write
never leaves a truncated chunk in the index; the index records each chunk's
``(sample_idx, question_idx)`` pairs so resuming does not re-read 1.6 GB chunks;
loaders shard by data-parallel rank and prefetch the next chunk into pinned
memory on a background thread.
"""
from __future__ import annotations

import json
import os
import queue
import threading
from datetime import datetime
from typing import Dict, Iterator, List, Optional, Sequence, Set, Tuple

import torch


def _load_pt(path: str) -> dict:
    return torch.load(path, map_location="cpu", weights_only=True)


class ChunkedHiddenStateWriter:
    def __init__(self, output_dir: str, chunk_size: int = 1000, config: Optional[dict] = None):
        self.output_dir = str(output_dir)
        self.chunks_dir = os.path.join(self.output_dir, "chunks")
        os.makedirs(self.chunks_dir, exist_ok=True)
        self.chunk_size = chunk_size
        self.config = config or {}
        self.index_path = os.path.join(self.output_dir, "index.json")
        self.total_samples = 0
        self.chunk_files: List[dict] = []
        self._egpt: List[torch.Tensor] = []
        self._vl: List[torch.Tensor] = []
        self._meta: List[dict] = []
        if os.path.exists(self.index_path):
            with open(self.index_path) as f:
                idx = json.load(f)
            self.total_samples = idx.get("total_samples", 0)
            self.chunk_files = idx.get("chunks", [])

    # -- write --------------------------------------------------------------
    def add_sample(self, egpt_hidden: torch.Tensor, vl_hidden: torch.Tensor, meta: dict,
                   source_chunk: Optional[str] = None) -> None:
        self._egpt.append(egpt_hidden.detach().to("cpu", torch.float32))
        self._vl.append(vl_hidden.detach().to("cpu", torch.float32))
        self._meta.append(dict(meta))
        self._source_chunk = source_chunk
        if len(self._egpt) >= self.chunk_size:
            self.flush()

    def flush(self) -> Optional[str]:
        if not self._egpt:
            return None
        n = len(self._egpt)
        chunk_id = self.total_samples
        max_seq = max(max(h.shape[0] for h in self._egpt), max(h.shape[0] for h in self._vl))
        hd = self._egpt[0].shape[1]
        e = torch.zeros(n, max_seq, hd)
        v = torch.zeros(n, max_seq, hd)
        lens = []
        for i, (eh, vh) in enumerate(zip(self._egpt, self._vl)):
            e[i, :eh.shape[0]] = eh
            v[i, :vh.shape[0]] = vh
            lens.append(min(eh.shape[0], vh.shape[0]))
        name = f"chunk_{chunk_id:06d}.pt"
        final = os.path.join(self.chunks_dir, name)
        tmp = final + ".tmp"
        torch.save({"egpt_hidden": e, "vl_hidden": v, "seq_lens": torch.tensor(lens),
                    "metadata": self._meta, "chunk_id": chunk_id, "n_samples": n}, tmp)
        os.replace(tmp, final)
        entry = {"path": name, "start_idx": chunk_id, "n_samples": n,
                 "pairs": [[m.get("sample_idx", -1), m.get("question_idx", -1)] for m in self._meta]}
        if getattr(self, "_source_chunk", None):
            entry["source_chunk"] = self._source_chunk
        self.chunk_files.append(entry)
        self.total_samples += n
        self._egpt, self._vl, self._meta = [], [], []
        self._save_index()
        return final

    def _save_index(self) -> None:
        tmp = self.index_path + ".tmp"
        with open(tmp, "w") as f:
            json.dump({"total_samples": self.total_samples, "chunk_size": self.chunk_size,
                       "chunks": self.chunk_files, "config": self.config,
                       "created": datetime.now().isoformat()}, f, indent=2)
        os.replace(tmp, self.index_path)

    # -- resume -------------------------------------------------------------
    def get_processed_pairs(self) -> Set[Tuple[int, int]]:
        done: Set[Tuple[int, int]] = set()
        for info in self.chunk_files:
            if "pairs" in info:
                done.update((int(a), int(b)) for a, b in info["pairs"])
                continue
            path = os.path.join(self.chunks_dir, info["path"])
            if os.path.exists(path):
                for m in _load_pt(path)["metadata"]:
                    done.add((m["sample_idx"], m["question_idx"]))
        return done

    def processed_source_chunks(self) -> Set[str]:
        return {c["source_chunk"] for c in self.chunk_files if "source_chunk" in c}

    @property
    def pending(self) -> int:
        return len(self._egpt)


def read_index(data_dir: str) -> dict:
    path = os.path.join(data_dir, "index.json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Index not found: {path}")
    with open(path) as f:
        return json.load(f)


def iter_chunks(data_dir: str) -> Iterator[dict]:
    idx = read_index(data_dir)
    for info in idx["chunks"]:
        yield _load_pt(os.path.join(data_dir, "chunks", info["path"]))


def load_chunked_hidden_states(data_dir: str) -> dict:
    """Concatenate every chunk into ``[N, max_seq, H]`` tensors (small sets only)."""
    idx = read_index(data_dir)
    chunks = list(iter_chunks(data_dir))
    if not chunks:
        raise ValueError(f"{data_dir}: no chunks")
    max_seq = max(c["egpt_hidden"].shape[1] for c in chunks)
    hd = chunks[0]["egpt_hidden"].shape[2]
    n = sum(c["egpt_hidden"].shape[0] for c in chunks)
    e, v = torch.zeros(n, max_seq, hd), torch.zeros(n, max_seq, hd)
    meta, lens, at = [], [], 0
    for c in chunks:
        k, s = c["egpt_hidden"].shape[:2]
        e[at:at + k, :s] = c["egpt_hidden"]
        v[at:at + k, :s] = c["vl_hidden"]
        lens.append(c["seq_lens"])
        meta.extend(c["metadata"])
        at += k
    return {"egpt_hidden": e, "vl_hidden": v, "seq_lens": torch.cat(lens), "metadata": meta,
            "config": idx.get("config", {})}


def load_hidden_states(path: str) -> dict:
    """A chunked directory or a single ``.pt`` with the same keys."""
    return load_chunked_hidden_states(path) if os.path.isdir(path) else _load_pt(path)


# ------------------------------------------------------------------- batching
def make_batch(egpt: torch.Tensor, vl: torch.Tensor, seq_lens: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Trim to the batch's longest sequence and build the validity mask.  ``seq_lens`` is a host
    tensor; ``egpt`` / ``vl`` may live on a device, in which case the mask is built there."""
    longest = int(seq_lens.max())
    dev = egpt.device
    lens = seq_lens.to(dev) if dev.type != "cpu" else seq_lens
    mask = (torch.arange(longest, device=dev)[None, :] < lens[:, None]).float()
    return {"egpt_hidden": egpt[:, :longest], "vl_hidden": vl[:, :longest], "mask": mask,
            "seq_lens": seq_lens}


def _take(t: torch.Tensor, sel: torch.Tensor) -> torch.Tensor:
    """Rows ``sel`` (host index tensor) of ``t`` on ``t``'s device."""
    return t.index_select(0, sel.to(t.device)) if t.device.type != "cpu" else t[sel]


class DeviceBatchLoader:
    """Mini-batches cut on the accelerator from a dataset that is resident there.

    The host-side alternative gathers ``batch x seq x 4096`` fp32 rows twice per step (52 MB
    each at batch 64) and uploads them from pageable memory, which bounds L1-L4 adapter
    training at a third of what the GPU sustains; here a step uploads ``batch`` indices."""

    def __init__(self, data: dict, indices: Optional[Sequence[int]], batch_size: int, device,
                 shuffle: bool = True, seed: int = 0, drop_last: bool = False):
        self.device = torch.device(device)
        idx = torch.arange(data["egpt_hidden"].shape[0]) if indices is None else torch.as_tensor(list(indices))
        self.lens = data["seq_lens"][idx].clone()
        self.e = data["egpt_hidden"][idx].to(self.device)
        self.v = self.e if data["vl_hidden"] is data["egpt_hidden"] else data["vl_hidden"][idx].to(self.device)
        self.batch_size, self.shuffle, self.seed, self.drop_last = batch_size, shuffle, seed, drop_last
        self.epoch = 0

    @staticmethod
    def fits(data: dict, device, fraction: float = 0.5) -> bool:
        device = torch.device(device)
        if device.type != "cuda":
            return False
        need = sum(data[k].numel() * data[k].element_size() for k in ("egpt_hidden", "vl_hidden"))
        free, _ = torch.cuda.mem_get_info(device)
        return need < fraction * free

    def __len__(self) -> int:
        n = self.lens.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        n = self.lens.shape[0]
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        perm = torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)
        for s in range(0, n, self.batch_size):
            sel = perm[s:s + self.batch_size]
            if self.drop_last and len(sel) < self.batch_size:
                break
            yield make_batch(_take(self.e, sel), _take(self.v, sel), self.lens[sel])
        self.epoch += 1


def collate_fn(items: Sequence[dict]) -> Dict[str, torch.Tensor]:
    """Pad variable-length samples ``{egpt_hidden [S,H], vl_hidden [S,H]}`` to the batch max."""
    lens = torch.tensor([min(it["egpt_hidden"].shape[0], it["vl_hidden"].shape[0]) for it in items])
    longest, hd = int(lens.max()), items[0]["egpt_hidden"].shape[1]
    e = torch.zeros(len(items), longest, hd)
    v = torch.zeros(len(items), longest, hd)
    for i, it in enumerate(items):
        n = int(lens[i])
        e[i, :n] = it["egpt_hidden"][:n]
        v[i, :n] = it["vl_hidden"][:n]
    return make_batch(e, v, lens)


class HiddenStateDataset(torch.utils.data.Dataset):
    """In-memory dataset over ``{egpt_hidden, vl_hidden, seq_lens}`` tensors."""

    def __init__(self, data: dict, indices: Optional[Sequence[int]] = None):
        self.e, self.v, self.lens = data["egpt_hidden"], data["vl_hidden"], data["seq_lens"]
        self.indices = list(range(self.e.shape[0])) if indices is None else list(indices)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        j = self.indices[i]
        n = int(self.lens[j])
        return {"egpt_hidden": self.e[j, :n], "vl_hidden": self.v[j, :n]}


class _ChunkStream:
    """Background reader: yields chunk dicts in a given order, one chunk ahead."""

    def __init__(self, data_dir: str, order: Sequence[int], pin: bool, device=None):
        self.data_dir, self.order, self.pin = data_dir, list(order), pin
        self.device = torch.device(device) if device is not None else None
        self.side = torch.cuda.Stream(self.device) if self.device is not None and self.device.type == "cuda" else None
        self.index = read_index(data_dir)
        self.q: "queue.Queue" = queue.Queue(maxsize=1)
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        try:
            for ci in self.order:
                info = self.index["chunks"][ci]
                d = _load_pt(os.path.join(self.data_dir, "chunks", info["path"]))
                if self.pin and torch.cuda.is_available():
                    for k in ("egpt_hidden", "vl_hidden"):
                        d[k] = d[k].pin_memory()
                if self.side is not None:
                    # whole chunk to the accelerator on a side stream, while the consumer trains on
                    # the previous one; batches are then cut on the device
                    with torch.cuda.stream(self.side):
                        for k in ("egpt_hidden", "vl_hidden"):
                            d[k] = d[k].to(self.device, non_blocking=True)
                        d["_ready"] = self.side.record_event()
                self.q.put(d)
        except BaseException as exc:  # surfaced in the consumer thread
            self.q.put(exc)
        self.q.put(None)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            ev = item.pop("_ready", None)
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for k in ("egpt_hidden", "vl_hidden"):   # freed blocks wait for the consumer's kernels
                    item[k].record_stream(cur)
            yield item


class ChunkedTrainLoader:
    """Shuffled mini-batches streamed chunk by chunk.

    Chunk order and the in-chunk permutation are reshuffled each epoch from
    ``seed + epoch`` (identical on every rank); rank ``r`` of ``world`` takes every
    ``world``-th sample of each chunk permutation, so ranks see disjoint data and
    the same number of batches (trailing samples that do not fill a batch on all
    ranks are dropped).
    """

    def __init__(self, data_dir: str, batch_size: int = 64, shuffle: bool = True, seed: int = 0,
                 rank: int = 0, world: int = 1, pin: bool = True, drop_last: bool = False, device=None):
        self.data_dir, self.batch_size, self.shuffle = data_dir, batch_size, shuffle
        self.seed, self.rank, self.world, self.pin = seed, rank, world, pin
        self.drop_last = drop_last
        self.device = device      # chunks are uploaded whole and batches cut on this device
        self.index = read_index(data_dir)
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _per_rank(self, n: int) -> int:
        return n // self.world if self.world > 1 else n

    def __len__(self) -> int:
        total = 0
        for c in self.index["chunks"]:
            k = self._per_rank(c["n_samples"])
            total += k // self.batch_size if (self.drop_last or self.world > 1) \
                else (k + self.batch_size - 1) // self.batch_size
        return total

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        n_chunks = len(self.index["chunks"])
        order = torch.randperm(n_chunks, generator=g).tolist() if self.shuffle else list(range(n_chunks))
        for chunk in _ChunkStream(self.data_dir, order, self.pin, self.device):
            n = chunk["egpt_hidden"].shape[0]
            perm = torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)
            if self.world > 1:
                perm = perm[: (n // self.world) * self.world][self.rank::self.world]
            full_only = self.drop_last or self.world > 1
            for s in range(0, len(perm), self.batch_size):
                sel = perm[s:s + self.batch_size]
                if full_only and len(sel) < self.batch_size:
                    break
                yield make_batch(_take(chunk["egpt_hidden"], sel), _take(chunk["vl_hidden"], sel),
                                 chunk["seq_lens"][sel])
        self.epoch += 1


class ChunkedValLoader(ChunkedTrainLoader):
    """Sequential pass over every sample of every chunk (all batches are yielded —
    the reference's loader returned only the last batch of each chunk,
    pipeline/adapter_train/train_hidden_adapter.py:227-242)."""

    def __init__(self, data_dir: str, batch_size: int = 64, rank: int = 0, world: int = 1,
                 pin: bool = True, device=None):
        super().__init__(data_dir, batch_size, shuffle=False, rank=rank, world=world, pin=pin, device=device)

    def __len__(self) -> int:
        return sum((self._shard(c["n_samples"]) + self.batch_size - 1) // self.batch_size
                   for c in self.index["chunks"])

    def _shard(self, n: int) -> int:
        return len(range(self.rank, n, self.world))

    def __iter__(self):
        order = list(range(len(self.index["chunks"])))
        for chunk in _ChunkStream(self.data_dir, order, self.pin, self.device):
            idx = torch.arange(chunk["egpt_hidden"].shape[0])[self.rank::self.world]
            for s in range(0, len(idx), self.batch_size):
                sel = idx[s:s + self.batch_size]
                yield make_batch(_take(chunk["egpt_hidden"], sel), _take(chunk["vl_hidden"], sel),
                                 chunk["seq_lens"][sel])
