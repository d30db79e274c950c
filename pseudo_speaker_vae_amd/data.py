"""PackedEmbeddingTable: the whole dataset as ONE [N, D] table in HBM plus on-device batch assembly.

The reference's datasets load one tiny .pth/.pt file per sample inside __getitem__
(/root/reference/ps_vae/data/cv.py:58-76, vctk.py:51-67), which makes real-data training I/O-bound on
674,068 files (plots/dataset_info_train.json).  On MI355X the table (674,068 x 256 x 4 B = 690 MB) fits
HBM 400 times over: pack once, then every batch is an index gather at HBM speed (psvae_gather_rows_f32).
Labels are the integer indices of labels.py (utils.py:82-136)."""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional, Tuple

import torch

from . import _ffi


class PackedEmbeddingTable:
    def __init__(self, embeddings: torch.Tensor, labels: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0"):
        if embeddings.dim() != 2:
            raise ValueError("embeddings must be [N, D]")
        self.device = torch.device(device)
        self.table = embeddings.to(device=self.device, dtype=torch.float32).contiguous()
        self.labels = {k: v.to(self.device) for k, v in (labels or {}).items()}
        for k, v in self.labels.items():
            if v.shape[0] != self.table.shape[0]:
                raise ValueError(f"label {k}: {v.shape[0]} entries for {self.table.shape[0]} embeddings")

    @classmethod
    def from_samples(cls, samples: Iterable[torch.Tensor], labels=None, device="cuda:0"):
        """samples: per-utterance embeddings of any shape with D elements (FreeVC stores (1, D, 1))"""
        return cls(torch.stack([s.reshape(-1).float() for s in samples]), labels, device)

    def __len__(self):
        return self.table.shape[0]

    @property
    def dim(self):
        return self.table.shape[1]

    def batch(self, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        idx = idx.to(device=self.device, dtype=torch.int64).contiguous()
        B = idx.shape[0]
        out = out if out is not None else torch.empty(B, self.dim, dtype=torch.float32, device=self.device)
        _ffi.check(_ffi.lib().psvae_gather_rows_f32(self.table.data_ptr(), len(self), self.dim, idx.data_ptr(), B,
                                                    out.data_ptr(), _ffi.stream_ptr()), "gather_rows")
        return out

    def epoch(self, batch_size: int, shuffle: bool = True, drop_last: bool = False,
              generator: Optional[torch.Generator] = None) -> Iterator[Tuple[torch.Tensor, Dict[str, torch.Tensor]]]:
        """(x, labels) batches; the permutation is drawn on the host, everything else stays on the device"""
        n = len(self)
        perm = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        perm = perm.to(self.device)
        for s in range(0, n, batch_size):
            idx = perm[s:s + batch_size]
            if drop_last and idx.shape[0] < batch_size:
                return
            yield self.batch(idx), {k: v[idx] for k, v in self.labels.items()}
