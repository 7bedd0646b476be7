"""The touched-set exchange's device pieces alone (sls_exchange.hip): the OR of the ranks' bitmaps,
its exclusive bit-count prefix, the union's size and the group's verdict — against NumPy, at sizes that take every path of
the one-workgroup kernel (runs of odd and even length in LDS, the padded layout, models too large for LDS)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,G", [(1, 1), (63, 2), (5000, 2), (65600, 3), (100000, 2), (500000, 8), (786432, 1), (1000001, 3)])
def test_grad_union_matches_numpy(device, N, G):
    from splat_loam_amd import _abi
    lib = _abi.lib()
    rng = np.random.default_rng(N + G)
    nw = int(lib.sls_grad_bitmap_words(N))
    assert nw == (N + 63) // 64 + 2
    maps = np.zeros((G, nw), np.uint64)
    for g in range(G):
        bits = rng.random(nw * 64 - 128) < (0.02 + 0.1 * g)
        bits[N:] = False                                     # (bits beyond N are never set by the producers)
        maps[g, :nw - 2] = np.packbits(bits, bitorder="little").view(np.uint64)
    maps[G - 1, nw - 1] = 1 if N % 2 else 0                  # one rank voids the iteration (bit 1 of the verdict) for odd N
    want = np.bitwise_or.reduce(maps, axis=0)
    counts = np.array([bin(int(w)).count("1") for w in want[:nw - 2]], np.int64)
    prefix = np.concatenate([[0], np.cumsum(counts)[:-1]])
    K = int(counts.sum())
    for cap in (max(K, 1), max(K - 1, 1)):
        d_maps = torch.tensor(maps.view(np.int64).reshape(-1), device=device)
        d_union = torch.full((nw,), -1, dtype=torch.int64, device=device)
        d_prefix = torch.full((nw,), -1, dtype=torch.int32, device=device)
        status = torch.zeros((8,), dtype=torch.int32, device=device)
        _abi.check(lib.sls_grad_union(N, d_maps.data_ptr(), G, d_union.data_ptr(), cap, d_prefix.data_ptr(), status.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "sls_grad_union")
        torch.cuda.synchronize()
        assert np.array_equal(d_union.cpu().numpy().view(np.uint64), want)
        assert np.array_equal(d_prefix.cpu().numpy()[:nw - 2].astype(np.int64), prefix)
        st = status.cpu().numpy()
        assert int(st[7]) == K
        assert int(st[1]) == (2 if N % 2 else 0) | (4 if K > cap else 0)
