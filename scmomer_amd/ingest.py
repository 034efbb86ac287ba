"""Cell ingest on the device: the expression matrix stays resident in HBM as CSR and a batch of cells is gathered,
densified and binned to tokens by one kernel (no per-cell host work, no host->device copy per step).

Mirrors SCDataset of the S1 script (s1_pretrain_rna.py:127-140: ``data[index].toarray()[0]``, values above
``class_limit - 2`` set to it, ``.long()``) and the binning inside scMomer.forward / get_cell_embedding
(scmomer.py:56,108).  Values are held as float32 (what the reference's matrices hold); integer results are
bit-identical to the reference's for float32 input, and float64 input that would bin differently is refused.
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import ops

RULE_S1, RULE_SCMOMER = ops.RULE_S1, ops.RULE_SCMOMER


class DeviceCSR:
    """CSR matrix [n_cells, n_genes] in device memory: indptr int64, indices int32, values f32."""

    def __init__(self, indptr, indices, values, n_genes: int, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise ValueError("DeviceCSR keeps the matrix in GPU memory; pass a cuda device")
        indptr = np.ascontiguousarray(np.asarray(indptr), dtype=np.int64)
        indices = np.ascontiguousarray(np.asarray(indices), dtype=np.int32)
        values = np.ascontiguousarray(np.asarray(values), dtype=np.float32)
        if indptr.ndim != 1 or indptr.size < 1 or indptr[0] != 0 or indptr[-1] != indices.size \
                or indices.size != values.size or np.any(np.diff(indptr) < 0):
            raise ValueError("malformed CSR arrays")
        if indices.size and (indices.min() < 0 or indices.max() >= n_genes):
            raise ValueError(f"column index outside [0, {n_genes})")
        self.n_cells = indptr.size - 1
        self.n_genes = int(n_genes)
        self.indptr = torch.from_numpy(indptr).to(device)
        # one spare entry so that an empty matrix still has valid device pointers
        self.indices = torch.from_numpy(np.append(indices, np.int32(0))).to(device)
        self.values = torch.from_numpy(np.append(values, np.float32(0))).to(device)
        self.err_flag = torch.zeros((), dtype=torch.int32, device=device)
        self.device = device

    @classmethod
    def from_scipy(cls, m, device) -> "DeviceCSR":
        """m: anything scipy.sparse can turn into CSR (the ``.X`` of an AnnData), or a dense 2-D array."""
        import scipy.sparse as sp
        m = sp.csr_matrix(m)
        if m.data.dtype == np.float64 and m.data.size:
            # The matrix is kept in float32 on the device, while the reference bins in the matrix's own dtype
            # (s1_pretrain_rna.py:134-136).  The two agree unless rounding a float64 value to float32 moves it across
            # an integer (e.g. 2.9999999999 -> 3.0): refuse such a matrix instead of binning a cell differently.
            d64 = m.data
            d32 = d64.astype(np.float32).astype(np.float64)
            if np.any(np.trunc(d64) != np.trunc(d32)) or np.any(np.rint(d64) != np.rint(d32)):
                raise ValueError("float64 expression values whose float32 rounding crosses an integer boundary: "
                                 "bins would differ from the reference's; convert the matrix to float32 first")
        return cls(m.indptr, m.indices, m.data, m.shape[1], device)

    @property
    def shape(self):
        return (self.n_cells, self.n_genes)

    def tokens(self, rows: Union[torch.Tensor, Sequence[int], None], rule: int = RULE_S1, limit: int = 5,
               batch: Optional[int] = None) -> torch.Tensor:
        """int64 [B, n_genes] tokens of the listed cells (rows None: cells 0..batch-1)."""
        if rows is not None and not torch.is_tensor(rows):
            rows = torch.as_tensor(list(rows), dtype=torch.int64)
        if rows is not None:
            rows = rows.to(device=self.device, dtype=torch.int64).contiguous()
        return ops.bin_csr(self.indptr, self.indices, self.values, rows, self.n_genes, rule, limit, self.err_flag,
                           batch=batch)

    def raise_on_bad_index(self) -> None:
        """Synchronising check of the flag the kernel sets for row/column numbers outside the matrix."""
        flag = int(self.err_flag.item())
        if flag:
            self.err_flag.zero_()
            raise IndexError("row number outside the matrix" if flag & 1 else "column index outside the gene range")


class SCDataset:
    """SCDataset(data, class_limit) of s1_pretrain_rna.py:127-140 with the matrix on the device.  ``dataset[i]`` is
    the int64 token row of cell i (a device tensor); ``dataset.batch(indices)`` bins a whole batch in one launch."""

    def __init__(self, data, class_limit: int, device=None):
        if isinstance(data, DeviceCSR):
            self.data = data
        else:
            self.data = DeviceCSR.from_scipy(data, device if device is not None else torch.device("cuda"))
        self.class_limit = class_limit

    def __getitem__(self, index: int) -> torch.Tensor:
        n = len(self)
        if not -n <= index < n:
            raise IndexError(index)
        return self.batch([index % n])[0]

    def batch(self, indices) -> torch.Tensor:
        return self.data.tokens(indices, RULE_S1, self.class_limit - 2)

    def __len__(self) -> int:
        return self.data.n_cells


def bin_cells(x: torch.Tensor, rule: int = RULE_SCMOMER, limit: int = 5) -> torch.Tensor:
    """Dense expression values (any shape, device f32) -> int64 tokens with the S1 or the scMomer rule."""
    return ops.bin_dense(x.to(torch.float32).contiguous(), rule, limit)
