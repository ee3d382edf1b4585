"""The entry point that allocates nothing (qk_kernel_stats_noalloc_dev: partial sums in a static device
array inside the library) against the existing one (qk_kernel_stats_dev: caller workspace), through the C
ABI, bitwise, on two calls in a row -- the second on a smaller matrix, so anything the first call left in
the shared array would show."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(lib, _lib, torch, K, labels):
    N = K.shape[0]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    want = torch.full((_lib.STATS_LEN,), -7.0, dtype=torch.float64, device="cuda")
    ws = torch.empty(_lib.STATS_WS, dtype=torch.float64, device="cuda")
    _lib.check(lib.qk_kernel_stats_dev(P(K), N, K.stride(0), P(labels), P(want), P(ws), stream))
    got = torch.full((_lib.STATS_LEN,), -9.0, dtype=torch.float64, device="cuda")
    _lib.check(lib.qk_kernel_stats_noalloc_dev(P(K), N, K.stride(0), P(labels), P(got), stream))
    torch.cuda.synchronize()
    return got.cpu().numpy(), want.cpu().numpy()


def test_noalloc_stats_match_workspace_stats_twice():
    import torch
    from quake_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    for N, with_labels in [(1500, True), (333, True), (40, False), (2100, True)]:   # large, then smaller ones, then larger again
        A = rng.uniform(0, 1, size=(N, N))
        Kh = (A + A.T) / 2
        np.fill_diagonal(Kh, 1.0)
        big = torch.zeros((N, N + 5), dtype=torch.float64, device="cuda")      # a row stride that is not N
        K = big[:, :N]
        K.copy_(torch.from_numpy(Kh))
        labels = torch.from_numpy(rng.integers(0, 2, size=N).astype(np.int32)).cuda() if with_labels else None
        got, want = _both(lib, _lib, torch, K, labels)
        assert np.array_equal(got, want), (N, got, want)                      # same kernels, same tree: bitwise
        iu = np.triu_indices(N, 1)
        assert abs(got[0] - Kh[iu].sum()) < 1e-9 * max(1.0, Kh[iu].sum())
        assert abs(got[1] - (Kh[iu] ** 2).sum()) < 1e-9 * max(1.0, (Kh[iu] ** 2).sum())


def test_noalloc_stats_rejects_oversize_and_bad_arguments():
    import torch
    from quake_amd import _lib
    lib = _lib.load()
    K = torch.ones((4, 4), dtype=torch.float64, device="cuda")
    out = torch.zeros(_lib.STATS_LEN, dtype=torch.float64, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.qk_kernel_stats_noalloc_dev(P(K), 131073, 131073, None, P(out), None)
    assert rc == _lib.QK_ERR_UNSUPPORTED and "QK_STATS_NOALLOC_MAX_N" in _lib.last_error()
    assert lib.qk_kernel_stats_noalloc_dev(None, 4, 4, None, P(out), None) == _lib.QK_ERR_INVALID
    assert lib.qk_kernel_stats_noalloc_dev(P(K), 4, 3, None, P(out), None) == _lib.QK_ERR_INVALID
