"""The workgroup -> (problem, query tile) maps of the k-NN kernels only place work (gkg_knn_common.h: knn_map): a call with
B*G >= 16 problems and a positional bias of >= 8 MB takes the interleaved map (groups of an XCD's problems adjacent per
query tile, so that the bias rows are fetched once per group); its graphs must be bit-identical to the same problems run
two at a time (B*G = 2: problem-major map), for the fp32 tile kernel, the prefilter kernel (whose clean-up pass reads
per-workgroup flags through the same map) and a group count that is not a power of two."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("BG,c,N,M,k,d,prefilter", [
    (16, 24, 2304, 1296, 9, 1, "0"),        # fp32 tile kernel, 2 problems per XCD
    (32, 40, 1700, 1296, 9, 1, "force"),    # prefilter kernel (+ clean-up pass), 4 per XCD, ragged query tiles
    (24, 16, 2304, 1024, 18, 1, "0"),       # 3 problems per XCD (odd group), longer lists
    (64, 40, 2048, 1296, 9, 2, "force"),    # 8 per XCD, dilation 2
])
def test_interleaved_map_gives_the_same_graphs(BG, c, N, M, k, d, prefilter, monkeypatch):
    from gkgnet_amd import ops
    monkeypatch.setenv("GKG_KNN_PREFILTER", prefilter)
    assert N * M * 4 >= 8 << 20
    gen = torch.Generator(device="cuda").manual_seed(BG * 7 + c)
    x = torch.randn(BG, c, N, device="cuda", generator=gen)
    y = torch.randn(BG, c, M, device="cuda", generator=gen)
    rp = -torch.rand(1, N, M, device="cuda", generator=gen)
    whole = ops.knn_graph(x, y, rp, k, d)[0]
    for b0 in range(0, BG, 2):
        pair = ops.knn_graph(x[b0:b0 + 2].contiguous(), y[b0:b0 + 2].contiguous(), rp, k, d)[0]
        assert torch.equal(whole[b0:b0 + 2], pair), f"problems {b0}, {b0 + 1}"
