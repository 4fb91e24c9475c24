"""On-disk dump of PQ code indices next to their labels, the input of the reference's offline
analyses (cross_entropy.py / edit_distance.py).  train.py:283-329 sketches it (commented out there):
indices nearest-upsampled to the label size, flattened to (b, H*W, num_pq), appended batch by batch
with NpyAppendArray.  Here: two plain ``.npy`` files that grow along axis 0 -- ``<prefix>_index.npy``
(rows, num_pq) int16 and ``<prefix>_label.npy`` (rows,) int16 -- readable with ``numpy.load`` (also
memory-mapped) after ``close()``.  int16 holds any codebook up to 32767 entries and labels from -1 up.
"""
import os

import numpy as np
import torch

from . import native

__all__ = ["CodeIndexDump"]

_HEADER_BYTES = 128   # fixed-size .npy v1.0 header so the shape can be rewritten in place


def _header(shape, dtype) -> bytes:
    d = "{'descr': '%s', 'fortran_order': False, 'shape': %s, }" % (np.dtype(dtype).str, repr(tuple(shape)))
    body = d.encode("latin1")
    pad = _HEADER_BYTES - 10 - len(body) - 1
    if pad < 0:
        raise ValueError("shape does not fit the fixed header")
    return b"\x93NUMPY\x01\x00" + (_HEADER_BYTES - 10).to_bytes(2, "little") + body + b" " * pad + b"\n"


class _GrowingNpy:
    def __init__(self, path, cols, dtype):
        self.path, self.cols, self.dtype, self.rows = path, cols, np.dtype(dtype), 0
        self.f = open(path, "wb")
        self.f.write(_header(self._shape(), self.dtype))

    def _shape(self):
        return (self.rows,) if self.cols is None else (self.rows, self.cols)

    def append(self, a: np.ndarray):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        self.f.write(a.tobytes())
        self.rows += a.shape[0]

    def close(self):
        self.f.seek(0)
        self.f.write(_header(self._shape(), self.dtype))
        self.f.close()


class CodeIndexDump:
    """``append(z_index, label)`` per batch, ``close()`` at the end (also a context manager).

    z_index: (num_pq, b, h, w) integer tensor as ``DIONPQGO.forward`` returns it; label: (b, H, W).
    With ``upsample=True`` (the reference's sketch) every label pixel gets the indices of the patch
    under it (nearest neighbour, as ``F.interpolate(mode="nearest")`` picks it), so both files have
    b*H*W rows per batch; otherwise the label is sampled at the same nearest positions the other way
    round and both have b*h*w rows."""

    def __init__(self, prefix: str, num_pq: int, upsample: bool = True):
        os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
        self.index = _GrowingNpy(prefix + "_index.npy", num_pq, np.int16)
        self.label = _GrowingNpy(prefix + "_label.npy", None, np.int16)
        self.num_pq, self.upsample = num_pq, upsample

    @torch.no_grad()
    def append(self, z_index: torch.Tensor, label: torch.Tensor):
        num_pq, b, h, w = z_index.shape
        if num_pq != self.num_pq:
            raise ValueError(f"expected {self.num_pq} sub-spaces, got {num_pq}")
        if int(z_index.max()) > 32767:
            raise ValueError("code index does not fit int16")
        hh, ww = label.shape[-2:]
        dev = z_index.device
        if z_index.is_cuda and label.is_cuda:
            # one HIP kernel writes the int16 rows directly (no up-sampled int64 intermediate)
            rows, lab = native.index_rows_i16(z_index.long(), label.long(), self.upsample)
            self.index.append(rows.cpu().numpy())
            self.label.append(lab.cpu().numpy())
            return
        if self.upsample:
            # nearest: source = floor(dst * in / out)   (ATen's nearest for integer scale ratios and beyond)
            ys = (torch.arange(hh, device=dev) * h) // hh
            xs = (torch.arange(ww, device=dev) * w) // ww
            idx = z_index[:, :, ys][:, :, :, xs]                       # (num_pq, b, H, W)
            lab = label
        else:
            ys = (torch.arange(h, device=label.device) * hh) // h
            xs = (torch.arange(w, device=label.device) * ww) // w
            idx = z_index
            lab = label[:, ys][:, :, xs]
        rows = idx.permute(1, 2, 3, 0).reshape(-1, num_pq).to(torch.int16).cpu().numpy()
        self.index.append(rows)
        self.label.append(lab.reshape(-1).to(torch.int16).cpu().numpy())

    def close(self):
        self.index.close()
        self.label.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
