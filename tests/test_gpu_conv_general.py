"""The general convolution kernels (any odd kernel size up to 7, any padding below it, stride 1 | 2, any channel
count: the reference's MyConv2d.forward, adp_qtz_fan.py:574-601, is written for any ksize / stride / padding) against
the CPU oracle, whose convolution is generic in the same parameters: forward, input gradient, weight gradient,
hard-mode codes and the shortcut addition.  Tolerance 2e-5 of the result's scale (fp32, other summation order).
Shapes: the ResNet-18 @ 224 layer shapes the 3x3 kernels of the ResNet-20 path cannot take (7x7 / stride 2 stem,
1x1 / stride 2 projections, 128-512 channels) at reduced batch, plus ragged cases."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import quantized_cnn_amd.ops as ops
DEV = "cuda"

SHAPES = [  # cin, cout, ks, stride, pad, H, W, N
    (3, 64, 7, 2, 3, 64, 64, 2),       # ResNet-18 stem at reduced size
    (3, 64, 7, 2, 3, 224, 224, 1),     # ... and at 224 x 224
    (3, 64, 7, 2, 3, 100, 120, 2),     # stem kernels: rows of 60 output pixels (four pixel tiles, the last one ragged), 50 rows
    (3, 128, 7, 2, 3, 50, 46, 3),      # ... two groups of 64 output channels; 23 pixels per row (no multiple of 4: general weight gradient)
    (3, 64, 7, 2, 2, 33, 37, 2),       # ... padding 2, odd sizes
    (64, 64, 3, 1, 1, 56, 56, 1),      # layer1: stride 1, padding 1, >= 64 channels either side run in the F(2x2, 3x3) form
    (64, 64, 3, 1, 1, 13, 9, 2),       # ... odd sizes: the last tile row / column owns one pixel row / column only
    (80, 96, 3, 1, 1, 7, 7, 3),        # ... a ragged group of output channels (96 = 64 + 32), 16 tiles covering 8 x 8 pixels
    (64, 64, 3, 1, 1, 5, 31, 1),       # ... wide and flat: blocks of 2 x 8 tiles
    (128, 64, 3, 1, 1, 1, 1, 4),       # ... a single pixel
    (64, 128, 3, 2, 1, 56, 56, 1),     # layer2.0.conv1
    (64, 128, 1, 2, 0, 56, 56, 2),     # layer2.0 projection
    (128, 128, 3, 1, 1, 28, 28, 2),
    (128, 256, 3, 2, 1, 28, 28, 1),
    (256, 256, 3, 1, 1, 14, 14, 2),
    (256, 512, 1, 2, 0, 14, 14, 2),
    (512, 512, 3, 1, 1, 7, 7, 3),
    (20, 24, 5, 1, 2, 9, 11, 2),       # 5x5, ragged everything
    (16, 16, 3, 1, 0, 10, 10, 2),      # no padding
    (5, 7, 3, 2, 2, 8, 8, 2),          # padding 2 on a 3x3
    (70, 130, 3, 2, 1, 13, 9, 2),      # channel counts that are no multiples of 16
    (8, 8, 1, 1, 0, 5, 5, 3),
    (33, 17, 7, 1, 3, 12, 12, 1),
    (32, 80, 3, 1, 0, 19, 23, 3),      # many-channel 3x3 kernel: no padding, 80 output channels (a ragged group of 64), odd sizes
    (48, 48, 3, 1, 2, 9, 9, 2),        # ... padding 2
    (16, 64, 3, 2, 1, 17, 15, 2),      # ... stride 2 on odd sizes
    (64, 64, 3, 1, 1, 40, 40, 1),      # ... 32-pixel-wide tiles with a ragged last column of tiles
    (64, 64, 3, 2, 2, 12, 10, 2),      # many-channel weight-gradient kernel: stride 2 with padding 2
    (32, 48, 3, 1, 1, 7, 7, 5),        # ... whole 7 x 7 images per tile, rows padded to 8 pixels, 48 output channels
    (64, 80, 3, 2, 1, 17, 15, 3),      # ... stride 2 on odd sizes, a ragged group of output channels
    (64, 64, 1, 1, 0, 9, 9, 2),        # ... 1x1 at stride 1
    (128, 48, 1, 2, 0, 11, 11, 3),     # ... 1x1 at stride 2 on odd sizes
    (48, 64, 3, 2, 0, 13, 14, 2),      # stride-2 input gradient by parity classes in the many-channel kernel: no padding
    (64, 96, 3, 2, 1, 30, 22, 2),      # ... padding 1, even sizes that are no multiples of the tile
]


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


def near(got, want, what, rel=2e-5):
    err = np.abs(got - want).max()
    assert err <= rel * max(np.abs(want).max(), 1e-6), (what, err, np.abs(want).max())


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_general_conv_forward_dgrad_wgrad_vs_oracle(shape):
    cin, cout, ks, stride, pad, H, Wd, N = shape
    rs = np.random.RandomState(sum(shape))
    x = rs.standard_normal((N, cin, H, Wd)).astype(np.float32)
    W = (rs.standard_normal((cout, cin, ks, ks)) / np.sqrt(cin * ks * ks)).astype(np.float32)
    y_ref = O.conv_fwd(x, W, stride, pad)
    xd, Wdv = cu(x), cu(W)
    y = ops.conv_forward(xd, Wdv, stride, pad=pad)
    assert tuple(y.shape) == y_ref.shape
    near(y.cpu().numpy(), y_ref, "y")
    res = rs.standard_normal(y_ref.shape).astype(np.float32)
    near(ops.conv_forward(xd, Wdv, stride, res=cu(res), pad=pad).cpu().numpy(), y_ref + res, "y + res")
    dy = rs.standard_normal(y_ref.shape).astype(np.float32)
    dyd = cu(dy)
    dx_ref = O.conv_dgrad(dy, W, x.shape, stride, pad)
    near(ops.conv_dgrad(dyd, Wdv, x.shape, stride, pad=pad).cpu().numpy(), dx_ref, "dx")
    xres = rs.standard_normal(x.shape).astype(np.float32)
    near(ops.conv_dgrad(dyd, Wdv, x.shape, stride, res=cu(xres), pad=pad).cpu().numpy(), dx_ref + xres, "dx + res")
    dW = ops.conv_wgrad(xd, dyd, stride, ks=ks, pad=pad)
    assert tuple(dW.shape) == W.shape
    near(dW.cpu().numpy(), O.conv_wgrad(x, dy, ks, stride, pad), "dW")
    # deterministic
    assert torch.equal(dW, ops.conv_wgrad(xd, dyd, stride, ks=ks, pad=pad))
    assert torch.equal(y, ops.conv_forward(xd, Wdv, stride, pad=pad))


@pytest.mark.parametrize("shape", [(3, 64, 7, 2, 3, 40, 40, 2), (128, 128, 3, 1, 1, 14, 14, 2), (64, 128, 1, 2, 0, 16, 16, 2)],
                         ids=lambda s: "x".join(map(str, s)))
def test_general_conv_from_codes(shape):
    """Hard mode (adp_qtz_fan_test.py:304-331): W = q[idx] decoded inside the kernel."""
    cin, cout, ks, stride, pad, H, Wd, N = shape
    rs = np.random.RandomState(7 + sum(shape))
    L = 32
    idx = rs.randint(0, L, (cout, cin, ks, ks)).astype(np.uint8)
    q = O.levels_hint(0.3, L)
    x = rs.standard_normal((N, cin, H, Wd)).astype(np.float32)
    y = ops.conv_forward_codes(cu(x), torch.from_numpy(idx).to(DEV), cu(q), stride, pad)
    near(y.cpu().numpy(), O.conv_fwd(x, q[idx.astype(np.int64)], stride, pad), "y(codes)")


def test_adjoint_identity_at_resnet18_sizes():
    """<conv(x, W), dy> = <x, dgrad(dy, W)> = <W, wgrad(x, dy)> at a full-size layer (batch 16, 256 -> 256 channels at
    14 x 14), where the oracle would take minutes: a size-independent property in float64 accumulation."""
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(16, 256, 14, 14, device=DEV, generator=g)
    W = torch.randn(256, 256, 3, 3, device=DEV, generator=g) * 0.02
    y = ops.conv_forward(x, W, 1)
    dy = torch.randn(y.shape, device=DEV, generator=g)
    a = float((y.double() * dy.double()).sum())
    b = float((x.double() * ops.conv_dgrad(dy, W, x.shape, 1).double()).sum())
    c = float((W.double() * ops.conv_wgrad(x, dy, 1).double()).sum())
    scale = float((y.double().abs() * dy.double().abs()).sum())
    assert abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale
    yt = torch.nn.functional.conv2d(x, W, padding=1)
    assert (y - yt).abs().max().item() <= 1e-4 * yt.abs().max().item()  # (stock fp32 conv may use other arithmetic)


ONE_BY_ONE = [  # cin, cout, stride, H, W, N: the GEMM kernel of the 1x1 layers (channel counts multiples of 16)
    (64, 128, 2, 56, 56, 2),     # ResNet-18 layer2.0 projection
    (128, 256, 2, 28, 28, 3),
    (256, 512, 2, 14, 14, 2),
    (64, 64, 1, 9, 9, 2),        # stride 1
    (48, 32, 2, 11, 13, 3),      # odd sizes: the last cells of the input gradient own a trailing row / column; 2 channel blocks per wavefront
    (32, 16, 2, 7, 6, 1),        # one channel block per wavefront; fewer than 16 pixels in the last wavefront
    (16, 80, 1, 5, 3, 5),
    (80, 48, 2, 2, 2, 70),       # one output pixel per image
]


@pytest.mark.parametrize("shape", ONE_BY_ONE, ids=lambda s: "x".join(map(str, s)))
def test_one_by_one_gemm_forward_and_dgrad_vs_oracle(shape):
    """1x1 / padding 0: forward (with and without the added tensor), input gradient (plain: every element of dx is
    written, the positions no output reads with zeros, into a buffer full of NaN; and with the added tensor)."""
    cin, cout, stride, H, Wd, N = shape
    rs = np.random.RandomState(11 + sum(shape))
    x = rs.standard_normal((N, cin, H, Wd)).astype(np.float32)
    W = (rs.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    y_ref = O.conv_fwd(x, W, stride, 0)
    xd, Wdv = cu(x), cu(W)
    near(ops.conv_forward(xd, Wdv, stride, pad=0).cpu().numpy(), y_ref, "y")
    res = rs.standard_normal(y_ref.shape).astype(np.float32)
    near(ops.conv_forward(xd, Wdv, stride, res=cu(res), pad=0).cpu().numpy(), y_ref + res, "y + res")
    dy = rs.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = O.conv_dgrad(dy, W, x.shape, stride, 0)
    dx = ops.conv_dgrad(cu(dy), Wdv, x.shape, stride, pad=0).cpu().numpy()
    assert np.isfinite(dx).all()
    near(dx, dx_ref, "dx")
    if stride == 2:  # exact zeros where no output reads x
        mask = np.ones((H, Wd), bool)
        mask[::2, ::2] = False
        assert np.all(dx[:, :, mask] == 0.0)
    xres = rs.standard_normal(x.shape).astype(np.float32)
    near(ops.conv_dgrad(cu(dy), Wdv, x.shape, stride, res=cu(xres), pad=0).cpu().numpy(), dx_ref + xres, "dx + res")


@pytest.mark.parametrize("shape", [(64, 64, 56, 4), (256, 256, 14, 4), (512, 512, 7, 4)], ids=lambda s: "x".join(map(str, s)))
def test_minimal_filtering_form_on_positive_activations(shape):
    """The F(2x2, 3x3) kernel subtracts neighbouring inputs before it multiplies: on activations as a network feeds them
    (after a ReLU: positive, far from zero mean) its forward and input-gradient results stay within 5e-6 of the result's
    scale of a float64 convolution (measured 4e-7 .. 1.1e-6; the stock fp32 operator: 4e-7 .. 7e-7)."""
    cin, cout, H, N = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(N, cin, H, H, device=DEV, generator=g).abs() + 3.0
    W = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (cin * 9) ** 0.5
    y64 = torch.nn.functional.conv2d(x.double(), W.double(), padding=1)
    y = ops.conv_forward(x, W, 1, pad=1)
    assert (y.double() - y64).abs().max().item() <= 5e-6 * y64.abs().max().item()
    dy = torch.randn(y.shape, device=DEV, generator=g).abs() + 1.0
    dx64 = torch.nn.functional.conv_transpose2d(dy.double(), W.double(), padding=1)
    dx = ops.conv_dgrad(dy, W, x.shape, 1, pad=1)
    assert (dx.double() - dx64).abs().max().item() <= 5e-6 * dx64.abs().max().item()
