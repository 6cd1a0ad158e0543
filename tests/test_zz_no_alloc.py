"""sempyr_gp_fwd_static — the one entry point that allocates nothing and takes no workspace (include/sempyr.h) — against
sempyr_gp_fwd with a caller-provided scratch buffer, through the C ABI: bit-identical norms and penalty on two calls in a
row, the second with a smaller input (the built-in scratch keeps the first call's contents), plus the size limit."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(L, grad):
    rows, n = grad.shape[0], grad[0].numel()
    dev = grad.device
    st = torch.cuda.current_stream(dev).cuda_stream
    norms_a = torch.empty(rows, device=dev); gp_a = torch.empty((), device=dev)
    norms_b = torch.full((rows,), -1.0, device=dev); gp_b = torch.full((), -1.0, device=dev)
    scratch = torch.empty(rows * 48, device=dev)
    assert L.sempyr_gp_fwd(grad.data_ptr(), rows, n, norms_a.data_ptr(), gp_a.data_ptr(), scratch.data_ptr(), st) == 0
    assert L.sempyr_gp_fwd_static(grad.data_ptr(), rows, n, norms_b.data_ptr(), gp_b.data_ptr(), st) == 0
    torch.cuda.synchronize()
    return norms_a, gp_a, norms_b, gp_b


def test_gp_fwd_static_matches_gp_fwd(dev):
    from semantic_pyramid_for_image_generation_amd import _lib
    L = _lib.lib
    g = torch.Generator().manual_seed(3)
    # 16 samples of 3x256x256 (the critic's input gradient at the bench batch: the split path that needs the scratch),
    # then a smaller call (4 samples), then a short-row call that takes the unsplit path
    for shape in ((16, 3, 256, 256), (4, 3, 256, 256), (5, 3, 16, 16)):
        grad = torch.randn(shape, generator=g).to(dev)
        na, ga, nb, gb = _both(L, grad)
        assert torch.equal(na, nb) and torch.equal(ga, gb), shape
        ref = grad.double().reshape(shape[0], -1).norm(dim=1)
        assert float((na.double() - ref).abs().max()) <= 1e-4 * float(ref.max())
        assert abs(float(ga) - float(((ref - 1) ** 2).mean())) <= 1e-4 * max(1.0, float(((ref - 1) ** 2).mean()))
    # beyond the built-in scratch: a clear error, nothing launched
    big = torch.zeros(257, 8, device=dev)
    out = torch.empty(257, device=dev); gp = torch.empty((), device=dev)
    rc = L.sempyr_gp_fwd_static(big.data_ptr(), 257, 8, out.data_ptr(), gp.data_ptr(), 0)
    assert rc < 0 and b'SEMPYR_GP_STATIC_MAX_ROWS' in L.sempyr_last_error_string()
