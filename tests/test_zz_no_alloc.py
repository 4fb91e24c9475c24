"""equss_pq_assign_static (the entry point that allocates nothing: its scratch is a buffer inside the
loaded library) against equss_pq_assign on the same inputs, through the C ABI: identical indices,
quantised rows and loss sums, on two calls in a row, the second with a smaller input, so that whatever
the first call left in the shared scratch cannot leak into the second."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(native, dev, rows, num_pq, k, d, seed, normalize="l2", gather_normalized=False):
    lib = native.load_library()
    rng = np.random.Generator(np.random.PCG64(seed))
    code = torch.from_numpy(rng.standard_normal((rows, num_pq * d)).astype(np.float32)).to(dev)
    books = torch.from_numpy(rng.uniform(-1.0 / k, 1.0 / k, size=(num_pq, k, d)).astype(np.float32)).to(dev)
    want_idx, want_zq, want_loss = native.pq_assign(code, books, normalize, gather_normalized)
    # outputs are allocated BEFORE the call under test; poisoned so that an unwritten element shows
    idx = torch.full((num_pq, rows), -7, dtype=torch.int64, device=dev)
    zq = torch.full((rows, num_pq * d), float("nan"), dtype=torch.float32, device=dev)
    loss = torch.full((num_pq,), float("nan"), dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    before = torch.cuda.memory_allocated(dev)
    rc = lib.equss_pq_assign_static(code.data_ptr(), code.stride(0), books.data_ptr(), idx.data_ptr(), zq.data_ptr(),
                                    num_pq * d, loss.data_ptr(), rows, num_pq, k, d, native.NORM_MODES[normalize],
                                    int(gather_normalized), stream)
    assert rc == 0, lib.equss_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(dev) == before
    assert torch.equal(idx, want_idx)
    assert torch.equal(zq, want_zq)
    assert torch.equal(loss, want_loss)


def test_static_scratch_entry_point_equals_workspace_entry_point(native, dev):
    lib = native.load_library()
    assert lib.equss_pq_static_scratch_bytes() >= lib.equss_pq_workspace_bytes(1, 16, 512, 128)
    _both(native, dev, rows=6400, num_pq=8, k=256, d=128, seed=41)        # screened kernel, S-cfg shape
    _both(native, dev, rows=1000, num_pq=4, k=128, d=128, seed=42)        # smaller input right after
    _both(native, dev, rows=777, num_pq=16, k=512, d=128, seed=43)        # B-cfg shape: the largest that fits
    _both(native, dev, rows=300, num_pq=8, k=64, d=32, seed=44, normalize="z_norm", gather_normalized=True)  # exact kernel


def test_static_scratch_entry_point_rejects_what_does_not_fit(native, dev):
    lib = native.load_library()
    rows, num_pq, k, d = 64, 32, 512, 128       # 8 MiB of prepared codebooks
    assert lib.equss_pq_workspace_bytes(rows, num_pq, k, d) > lib.equss_pq_static_scratch_bytes()
    code = torch.zeros(rows, num_pq * d, device=dev)
    books = torch.zeros(num_pq, k, d, device=dev)
    idx = torch.zeros(num_pq, rows, dtype=torch.int64, device=dev)
    rc = lib.equss_pq_assign_static(code.data_ptr(), code.stride(0), books.data_ptr(), idx.data_ptr(), None, num_pq * d,
                                    None, rows, num_pq, k, d, 1, 0, None)
    assert rc == 5   # EQUSS_E_WORKSPACE
    assert b"built-in buffer" in lib.equss_last_error()
