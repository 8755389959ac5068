"""Dataset container + ``.npz`` reader with the reference's file layout (main.py:780-820: keys
``node_features, edges, node_labels, node_texts, label_texts, train_masks, val_masks, test_masks``).

Only what feeds the hot path is here: tensors, texts, masks, the ``RandomState(seed)`` split of
main.py:792-808.  By default text arrays are read WITHOUT unpickling (``allow_pickle=False``): files that store texts as
fixed-width unicode arrays load directly; object-dtype (pickled) text arrays are refused with an explicit error
instead of executing a pickle.  ``allow_pickle=True`` is the reference's own behaviour (main.py:782) and is an explicit
opt-in for files the caller produced and trusts.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch


@dataclass
class GraphData:
    x: torch.Tensor
    edge_index: torch.Tensor
    y: torch.Tensor
    node_texts: List[str] = field(default_factory=list)
    label_texts: List[str] = field(default_factory=list)
    train_mask: Optional[torch.Tensor] = None
    val_mask: Optional[torch.Tensor] = None
    test_mask: Optional[torch.Tensor] = None

    @property
    def num_nodes(self) -> int:
        return self.x.size(0)

    def to(self, device):
        for k in ("x", "edge_index", "y", "train_mask", "val_mask", "test_mask"):
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, v.to(device))
        return self


def load_npz_dataset(npz_path: str, split_ratios: Optional[Tuple[float, float, float]] = None, seed: int = 42,
                     allow_pickle: bool = False):
    """-> (GraphData, num_features, num_classes), as ``load_npz_dataset`` in the reference (main.py:780-820)."""
    try:
        d = np.load(npz_path, allow_pickle=allow_pickle)
        texts = [str(s) for s in d["node_texts"]]
        label_texts = [str(s) for s in d["label_texts"]] if "label_texts" in d.files else []
    except ValueError as exc:
        raise ValueError(f"{npz_path}: text arrays are pickled object arrays; re-save them as unicode arrays "
                         "(np.array(texts, dtype=np.str_)), or pass allow_pickle=True for a file you trust") from exc
    x = torch.tensor(d["node_features"], dtype=torch.float)
    edge_index = torch.tensor(d["edges"], dtype=torch.long)
    y = torch.tensor(d["node_labels"], dtype=torch.long)
    n = x.size(0)
    if split_ratios is not None:
        train_ratio, val_ratio, _ = split_ratios
        idx = np.arange(n)
        np.random.RandomState(seed).shuffle(idx)
        n_train, n_val = int(train_ratio * n), int(val_ratio * n)
        masks = []
        for sel in (idx[:n_train], idx[n_train:n_train + n_val], idx[n_train + n_val:]):
            m = torch.zeros(n, dtype=torch.bool)
            m[sel] = True
            masks.append(m)
    else:
        masks = [torch.tensor(d[k], dtype=torch.bool) for k in ("train_masks", "val_masks", "test_masks")]
    data = GraphData(x, edge_index, y, texts, label_texts, *masks)
    return data, x.size(1), len(set(y.tolist()))
