"""The fp32 matrix-core GEMM of the dense training stages (gemm32.hip, ``dagl_gemm_f32``) against torch.matmul in fp64:
every operand layout, sizes that are not multiples of the 128 x 128 x 16 tile or of the 16-byte vector width, batches,
accumulation (beta) and the bias / relu epilogue."""
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("akc", [True, False])
@pytest.mark.parametrize("bkc", [True, False])
@pytest.mark.parametrize("nb,M,N,K", [(1, 120, 1710, 196), (2, 1710, 196, 120), (3, 37, 784, 1001), (1, 300, 130, 17)])
def test_gemm_f32_matches_fp64(akc, bkc, nb, M, N, K):
    from dagl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(nb, M, K, generator=g); Bm = torch.randn(nb, K, N, generator=g)
    want = A.double() @ Bm.double()
    a_st = A if akc else A.transpose(1, 2).contiguous()
    b_st = Bm.transpose(1, 2).contiguous() if bkc else Bm
    got = ops.gemm_f32(a_st.to(dev), b_st.to(dev), a_k_contiguous=akc, b_k_contiguous=bkc)
    assert got.shape == (nb, M, N)
    assert normwise(got.cpu().numpy(), want.numpy()) <= 2e-6
    # accumulate + epilogue
    C0 = torch.randn(nb, M, N, generator=g)
    bias = torch.randn(N, generator=g)
    out = C0.clone().to(dev)
    ops.gemm_f32(a_st.to(dev), b_st.to(dev), a_k_contiguous=akc, b_k_contiguous=bkc, out=out, alpha=0.5, beta=2.0,
                 bias=bias.to(dev), relu=True)
    want2 = torch.relu(0.5 * want + 2.0 * C0.double() + bias.double())
    assert normwise(out.cpu().numpy(), want2.numpy()) <= 2e-6


def test_gemm_f32_is_an_fmaf_chain_and_deterministic():
    """v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf chain: integer-valued operands give exact results, and two
    launches agree bit for bit."""
    from dagl_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-8, 9, (257, 333), generator=g).float()
    Bm = torch.randint(-8, 9, (333, 129), generator=g).float()
    got = ops.gemm_f32(A.to(dev), Bm.to(dev), True, False)
    assert torch.equal(got.cpu(), A @ Bm)
    x = torch.randn(500, 196, generator=g).to(dev); y = torch.randn(1234, 196, generator=g).to(dev)
    assert torch.equal(ops.gemm_f32(x, y, True, True), ops.gemm_f32(x, y, True, True))
