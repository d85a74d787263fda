"""PKT loss kernels (cosine-kernel matrix on the fp32 matrix cores, row-conditional probabilities, divergence and its
gradient) against a float64 restatement.  PARITY UNPINNED - no reference implementation: the reference cites the method
(README.md:24-29) but its training loops contain no teacher pass; oracle.pkt_loss is written from the published
method.  Tolerances: kernel matrix 2e-6 absolute (entries in [0, 1]); loss 2e-4 relative (a sum of terms that cancel
to second order, in fp32); gradient 2e-4 of its largest entry."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import quantized_cnn_amd.ops as ops
    from quantized_cnn_amd.pkt import pkt_loss
DEV = "cuda"


def ref_loss_and_grad(s, t, eps=1e-7):
    """The published formula in float64 with torch autograd on the CPU."""
    s = torch.from_numpy(s).double().requires_grad_(True)
    t = torch.from_numpy(t).double()

    def cond(x):
        xh = x / (torch.sqrt((x ** 2).sum(1, keepdim=True)) + eps)
        K = (xh @ xh.t() + 1.0) / 2.0
        return K / K.sum(1, keepdim=True)
    Ps, Pt = cond(s), cond(t)
    loss = torch.mean(Pt * torch.log((Pt + eps) / (Ps + eps)))
    loss.backward()
    return float(loss.detach()), s.grad.numpy()


@pytest.mark.parametrize("N,Ds,Dt", [(64, 16, 32), (100, 64, 10), (257, 130, 7), (1024, 64, 64), (1024, 64, 512)])
def test_pkt_loss_and_gradient_vs_float64(N, Ds, Dt):
    rs = np.random.RandomState(N + Ds)
    s = rs.standard_normal((N, Ds)).astype(np.float32)
    t = (rs.standard_normal((N, Dt)) + 0.3).astype(np.float32)
    want, gwant = ref_loss_and_grad(s, t)
    assert abs(want - O.pkt_loss(s, t)) <= 1e-12 + 1e-9 * abs(want)  # the numpy oracle and the torch restatement agree
    loss, d = ops.pkt_loss(torch.from_numpy(s).to(DEV), torch.from_numpy(t).to(DEV))
    assert abs(float(loss) - want) <= 2e-4 * abs(want) + 1e-10, (float(loss), want)
    err = np.abs(d.cpu().numpy() - gwant).max()
    assert err <= 2e-4 * np.abs(gwant).max(), (err, np.abs(gwant).max())
    loss2, d2 = ops.pkt_loss(torch.from_numpy(s).to(DEV), torch.from_numpy(t).to(DEV))
    assert torch.equal(loss, loss2) and torch.equal(d, d2)  # fixed-order sums: deterministic


@pytest.mark.parametrize("N,D", [(1024, 64), (333, 50), (16, 3)])
def test_pkt_kernel_matrix(N, D):
    rs = np.random.RandomState(N)
    x = rs.standard_normal((N, D)).astype(np.float32)
    x[N // 2] = 0.0  # a zero row: normalised to zeros, similarity 0 -> kernel value 1/2
    K = ops.pkt_kernel_matrix(torch.from_numpy(x).to(DEV)).cpu().numpy()
    xd = x.astype(np.float64)
    xh = xd / (np.sqrt((xd ** 2).sum(1, keepdims=True)) + 1e-7)
    want = (xh @ xh.T + 1.0) / 2.0
    assert np.abs(K - want).max() <= 2e-6
    assert np.array_equal(K, K.T)  # same k order on both sides of the diagonal: bitwise symmetric


def test_pkt_autograd_and_identical_features():
    torch.manual_seed(0)
    s = torch.randn(128, 64, device=DEV, requires_grad=True)
    t = torch.randn(128, 32, device=DEV)
    loss = pkt_loss(s, t)
    (3.0 * loss).backward()
    _, gwant = ref_loss_and_grad(s.detach().cpu().numpy(), t.cpu().numpy())
    assert np.abs(s.grad.cpu().numpy() - 3.0 * gwant).max() <= 2e-4 * 3.0 * np.abs(gwant).max()
    # a student that reproduces the teacher's similarities has zero divergence (scaling rows changes no cosine)
    same = pkt_loss((t * 2.5).clone().requires_grad_(True), t)
    assert abs(float(same)) <= 1e-9
    with pytest.raises(ValueError):
        ops.pkt_loss(torch.randn(4, 3, device=DEV), torch.randn(5, 3, device=DEV))
