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


@pytest.mark.parametrize("k,stride,C,H,W", [(7, 1, 16, 23, 30), (7, 4, 16, 23, 30), (3, 1, 64, 17, 21), (1, 1, 64, 9, 12),
                                            (7, 4, 64, 33, 47)])
def test_patch_linear_is_the_convolution_and_its_autograd(k, stride, C, H, W):
    """unfold + GEMM (+ bias, ReLU) and its explicit backward (train_ops.py) against torch's conv2d autograd in fp64:
    3x3 / 1x1 / 7x7 windows, stride 1 and the stride-4 SAME grid (asymmetric padding, dagl.py:123-139)."""
    import torch.nn.functional as F
    from dagl_amd import train_ops as T
    from dagl_amd.synth import same_pad_amounts
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(k * 100 + stride)
    B, O = 2, 20
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(O, C, k, k, generator=g) * 0.1
    b = torch.randn(O, generator=g)
    G = torch.randn(B, O, -(-H // stride), -(-W // stride), generator=g)
    # reference: SAME padding for stride 4, symmetric k//2 for stride 1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    if stride == 1:
        pt = pl = k // 2
        ref = F.relu(F.conv2d(xr, wr, br, padding=k // 2))
    else:
        (pt, pb), (pl, pr) = same_pad_amounts(H, k, stride), same_pad_amounts(W, k, stride)
        ref = F.relu(F.conv2d(F.pad(xr, (pl, pr, pt, pb)), wr, br, stride=stride))
    (ref * G.double()).sum().backward()
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    oh, ow = ref.shape[-2:]
    xp = T.to_padded_nhwc(xd, H, W)
    y = T.patch_linear(xp, T.conv_weight_rows(wd), bd, k, stride, T.PAD - pt, T.PAD - pl, oh, ow, relu=True)   # [B, oh*ow, O]
    out = y.view(B, oh, ow, O).permute(0, 3, 1, 2)
    assert normwise(out.detach().cpu().numpy(), ref.detach().numpy()) <= 2e-6
    (out * G.to(dev)).sum().backward()
    for got, want in ((xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        assert normwise(got.cpu().numpy(), want.numpy()) <= 5e-6


@pytest.mark.parametrize("B,H,W,stride,scale", [(2, 40, 36, 1, 1.0), (1, 64, 64, 1, 1e-6), (3, 23, 30, 4, 1e3), (8, 128, 128, 1, 1e-4)])
def test_fc_grad16_matches_fp64(B, H, W, stride, scale):
    """The split-fp16 gradient products of a patch projection (gemm16s.hip, ``dagl_fc_grad16``): d W = d Z^T rows and
    d rows = d Z W against fp64 -- stride-1 key grid and the stride-4 SAME query grid, gradients of very different
    magnitudes (the per-call power-of-two scale), the BASELINE config-5 size; two calls agree bit for bit."""
    import ctypes as C
    from dagl_amd import _lib, ops
    from dagl_amd.synth import same_pad_amounts
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + H)
    Hp, Wp = H + 6, W + 6
    pmap = torch.zeros(B, Hp, Wp, 16)
    pmap[:, 3:3 + H, 3:3 + W] = torch.randn(B, H, W, 16, generator=g)
    if stride == 1:
        oy = ox = 0; oh, ow = H, W
    else:
        t, l = same_pad_amounts(H, 7, 4)[0], same_pad_amounts(W, 7, 4)[0]
        oy, ox, oh, ow = 3 - t, 3 - l, -(-H // 4), -(-W // 4)
    n = B * oh * ow
    w = (torch.rand(196, 784, generator=g) - 0.5) * 0.07
    dz = torch.randn(n, 196, generator=g) * scale * torch.rand(n, 1, generator=g) ** 4          # heavy-tailed magnitudes
    y = torch.randn(n, 196, generator=g)                                                        # the layer's output: ReLU mask
    dy = dz
    dz = dy * (y > 0)
    pm, wd, dyd, yd = pmap.to(dev), w.to(dev), dy.to(dev), y.to(dev)
    dzd = dz.to(dev)
    need = lib.dagl_fc_grad16_scratch_bytes(B, oh, ow)
    scratch = torch.empty(need + 256, device=dev, dtype=torch.uint8)
    base = (scratch.data_ptr() + 255) // 256 * 256
    outs = []
    for rep in range(3):                      # 0, 1: ReLU backward fused (y given); 2: d z handed in directly -- all bit-identical
        d_w = torch.empty(196, 784, device=dev); d_rows = torch.empty(n, 784, device=dev); d_b = torch.empty(196, device=dev)
        _lib.check(lib.dagl_fc_grad16(ops._stream(), B, Hp, Wp, stride, oy, ox, oh, ow, pm.data_ptr(), wd.data_ptr(),
                                      yd.data_ptr() if rep < 2 else None, dyd.data_ptr() if rep < 2 else dzd.data_ptr(),
                                      d_w.data_ptr(), d_b.data_ptr(), d_rows.data_ptr(), base, need), "dagl_fc_grad16")
        outs.append((d_w.cpu(), d_rows.cpu(), d_b.cpu()))
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    want_b = dz.double().sum(0)
    e_b = float((outs[0][2].double() - want_b).abs().max() / dz.double().abs().sum(0).max())
    assert e_b <= 1e-6, e_b
    # fp64 reference: rows by torch's unfold on the padded map (element order (kh,kw,c))
    rows = torch.empty(n, 784, dtype=torch.float64)
    x = pmap.double()
    idx = 0
    cols = []
    for kh in range(7):
        for kw in range(7):
            cols.append(x[:, oy + kh: oy + kh + (oh - 1) * stride + 1: stride, ox + kw: ox + kw + (ow - 1) * stride + 1: stride, :])
    rows = torch.stack(cols, dim=3).reshape(n, 784)                                              # [B,oh,ow,49,16]
    want_w = dz.double().t() @ rows
    want_r = dz.double() @ w.double()
    e_w, e_r = normwise(outs[0][0].numpy(), want_w.numpy()), normwise(outs[0][1].numpy(), want_r.numpy())
    print(f"[parity] fc_grad16 B={B} {H}x{W} stride {stride} |dz|~{scale:g}: d_w {e_w:.2e}, d_rows {e_r:.2e} (normwise vs fp64)")
    assert e_w <= 5e-6 and e_r <= 5e-6


@pytest.mark.parametrize("B,H,W,scale", [(8, 128, 128, 1e-4), (3, 40, 64, 1.0), (2, 24, 32, 1.0), (1, 16, 256, 1e3), (5, 33, 16, 1.0)])
def test_fc_grad16_dmap_folds_the_rows_inside_the_product(B, H, W, scale):
    """``dagl_fc_grad16_dmap`` (round 6): d map = fold(d Z W) of the stride-1 projection with the [n, 784] rows folded inside the product
    (over kw in the block's LDS, over kh by a second kernel) -- against fp64, against ``dagl_fc_grad16`` + ``dagl_fold_patches``, bit-equal
    across calls; widths of 16 / 32 / 64 patches (several image rows per row tile), 128 and 256 (two segments per row), the BASELINE
    config-5 size; a width the product cannot fold says so."""
    from dagl_amd import _lib, ops
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    Hp, Wp = H + 6, W + 6
    pmap = torch.zeros(B, Hp, Wp, 16)
    pmap[:, 3:3 + H, 3:3 + W] = torch.randn(B, H, W, 16, generator=g)
    n = B * H * W
    w = (torch.rand(196, 784, generator=g) - 0.5) * 0.07
    dy = torch.randn(n, 196, generator=g) * scale * torch.rand(n, 1, generator=g) ** 4
    y = torch.randn(n, 196, generator=g)
    dz = dy * (y > 0)
    assert lib.dagl_fc_grad16_dmap_ok(1, W) == 1 and lib.dagl_fc_grad16_dmap_ok(1, 72) == 0 and lib.dagl_fc_grad16_dmap_ok(4, W) == 0
    pm, wd, dyd, yd = pmap.to(dev), w.to(dev), dy.to(dev), y.to(dev)
    need = lib.dagl_fc_grad16_dmap_scratch_bytes(B, H, W)
    scratch = torch.empty(need + 256, device=dev, dtype=torch.uint8)
    base = (scratch.data_ptr() + 255) // 256 * 256
    outs = []
    for rep in range(2):
        d_w = torch.empty(196, 784, device=dev); d_b = torch.empty(196, device=dev); d_map = torch.full((B, Hp, Wp, 16), float("nan"), device=dev)
        _lib.check(lib.dagl_fc_grad16_dmap(ops._stream(), B, Hp, Wp, 1, 0, 0, H, W, pm.data_ptr(), wd.data_ptr(), yd.data_ptr(), dyd.data_ptr(),
                                           d_w.data_ptr(), d_b.data_ptr(), d_map.data_ptr(), base, need), "dagl_fc_grad16_dmap")
        outs.append((d_map.cpu(), d_w.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # the unfused pair on the library
    need2 = lib.dagl_fc_grad16_scratch_bytes(B, H, W)
    scratch2 = torch.empty(need2 + 256, device=dev, dtype=torch.uint8)
    base2 = (scratch2.data_ptr() + 255) // 256 * 256
    d_rows = torch.empty(n, 784, device=dev); d_w2 = torch.empty(196, 784, device=dev); d_map2 = torch.empty(B, Hp, Wp, 16, device=dev)
    _lib.check(lib.dagl_fc_grad16(ops._stream(), B, Hp, Wp, 1, 0, 0, H, W, pm.data_ptr(), wd.data_ptr(), yd.data_ptr(), dyd.data_ptr(),
                                  d_w2.data_ptr(), None, d_rows.data_ptr(), base2, need2), "dagl_fc_grad16")
    _lib.check(lib.dagl_fold_patches(ops._stream(), B, Hp, Wp, 16, 7, 1, 0, 0, H, W, d_rows.data_ptr(), d_map2.data_ptr()), "dagl_fold_patches")
    assert torch.equal(outs[0][1], d_w2.cpu())                                   # (d W does not know about the fold)
    # fp64: d map = fold(d z W)
    rows64 = (dz.double() @ w.double()).view(B, H, W, 7, 7, 16)
    want = torch.zeros(B, Hp, Wp, 16, dtype=torch.float64)
    for kh in range(7):
        for kw in range(7):
            want[:, kh:kh + H, kw:kw + W, :] += rows64[:, :, :, kh, kw, :]
    e = normwise(outs[0][0].numpy(), want.numpy())
    e2 = normwise(d_map2.cpu().numpy(), want.numpy())
    print(f"[parity] fc_grad16_dmap B={B} {H}x{W} |dz|~{scale:g}: folded in the product {e:.2e}, rows + fold {e2:.2e} (normwise vs fp64)")
    assert torch.isfinite(outs[0][0]).all() and e <= 5e-6


def test_projection_backward_fast_path_equals_the_fp32_gemm_path():
    """CE's fc2 layer under autograd: gradients through the split-fp16 gradient GEMM against the fp32 matrix-core GEMM path."""
    from dagl_amd import train_ops as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    B, H, W = 2, 32, 48
    res = {}
    for fast in (True, False):
        T.FAST_FC_BACKWARD = fast
        try:
            pmap = torch.zeros(B, H + 6, W + 6, 16)
            pmap[:, 3:3 + H, 3:3 + W] = torch.randn(B, H, W, 16, generator=torch.Generator().manual_seed(1))
            pm = pmap.to(dev).requires_grad_(True)
            w = ((torch.rand(196, 784, generator=torch.Generator().manual_seed(2)) - 0.5) * 0.07).to(dev).requires_grad_(True)
            b = torch.zeros(196, device=dev, requires_grad=True)
            y = T.patch_linear(pm, w, b, 7, 1, 0, 0, H, W, relu=True)
            (y * torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dev)).sum().backward()
            res[fast] = (pm.grad.cpu(), w.grad.cpu(), b.grad.cpu())
        finally:
            T.FAST_FC_BACKWARD = True
    for a, c in zip(res[True], res[False]):
        assert normwise(a.numpy(), c.numpy()) <= 1e-5


def test_prelu_on_the_library_matches_torch_forward_and_backward():
    """train_ops.PReLU (the ResBlocks' activation) against nn.PReLU: forward bit-equal, gradients equal up to the order of the
    slope gradient's sum (fp64 partials here), deterministic across runs; odd sizes / half precision take torch's path."""
    import torch
    from dagl_amd.train_ops import PReLU
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 40, 44, generator=g).to(dev)
    gy = torch.randn(2, 64, 40, 44, generator=g).to(dev)
    ours, ref = PReLU().to(dev), torch.nn.PReLU().to(dev)
    with torch.no_grad():
        ours.weight.fill_(0.2); ref.weight.fill_(0.2)
    outs = []
    for m in (ours, ref, ours):
        xi = x.clone().requires_grad_(True)
        m.weight.grad = None
        y = m(xi)
        y.backward(gy)
        outs.append((y.detach(), xi.grad.clone(), m.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref64 = float((gy.double() * x.double() * (x <= 0)).sum())
    assert abs(float(outs[0][2]) - ref64) <= 1e-6 * abs(ref64) + 1e-6
    assert abs(float(outs[1][2]) - ref64) <= 1e-4 * abs(ref64) + 1e-4          # (torch's own fp32 reduction, for scale)
    assert torch.equal(outs[0][2], outs[2][2])                                 # same bits on a repeated call
    assert sorted(ours.state_dict()) == sorted(ref.state_dict()) and ours.weight.shape == ref.weight.shape
    y_odd = ours(torch.randn(3, 5, 7, device=dev))                             # 105 elements: torch's path
    assert y_odd.shape == (3, 5, 7)


@pytest.mark.parametrize("B,H,W", [(2, 20, 24), (1, 37, 64), (3, 9, 4), (1, 5, 256), (8, 128, 128)])
def test_conv_pair_backward_on_the_maps_matches_fp64_autograd(B, H, W):
    """``dagl_conv_pair_backward`` (conv_grad.hip: the gradients of g 3x3 and theta 1x1, dagl.py:208-209, as tap-wise products
    on the maps) against torch's fp64 autograd of the same two convolutions, every output; the border of the padded gradient
    maps is filled with garbage (it belongs to constants of the forward and must not be read); same bits on a repeated call."""
    import torch.nn.functional as F
    from dagl_amd import _lib, ops
    from dagl_amd._lib import check
    dev = torch.device("cuda:0")
    lib = _lib.load()
    assert lib.dagl_conv_pair_backward_supported(B, H, W) == 1
    g = torch.Generator().manual_seed(H * 131 + W)
    x = torch.randn(B, 64, H, W, generator=g)
    gw = torch.randn(16, 64, 3, 3, generator=g) * 0.1
    tw = torch.randn(16, 64, 1, 1, generator=g) * 0.2
    d1 = torch.randn(B, H, W, 16, generator=g)
    d2 = torch.randn(B, H, W, 16, generator=g)
    # fp64 reference
    xr = x.double().requires_grad_(True); gwr = gw.double().requires_grad_(True); twr = tw.double().requires_grad_(True)
    gb = torch.zeros(16, dtype=torch.float64, requires_grad=True); tb = torch.zeros(16, dtype=torch.float64, requires_grad=True)
    b1 = F.conv2d(xr, gwr, gb, padding=1); b2 = F.conv2d(xr, twr, tb)
    (b1 * d1.double().permute(0, 3, 1, 2)).sum().add((b2 * d2.double().permute(0, 3, 1, 2)).sum()).backward()
    # padded NHWC gradient maps with a poisoned border
    def padded(d):
        p = torch.full((B, H + 6, W + 6, 16), 1e30)
        p[:, 3:3 + H, 3:3 + W] = d
        return p.to(dev)
    d1p, d2p = padded(d1), padded(d2)
    xd, gwd, twd = x.to(dev), gw.to(dev), tw.to(dev)
    outs = []
    for _ in range(2):
        dx = torch.empty(B, 64, H, W, device=dev); dgw = torch.empty(16, 64, 3, 3, device=dev); dgb = torch.empty(16, device=dev)
        dtw = torch.empty(16, 64, 1, 1, device=dev); dtb = torch.empty(16, device=dev)
        scr = torch.empty(max(16, lib.dagl_conv_pair_backward_scratch_bytes(B, H, W)), device=dev, dtype=torch.uint8)
        check(lib.dagl_conv_pair_backward(ops._stream(), B, H, W, xd.data_ptr(), d1p.data_ptr(), d2p.data_ptr(), gwd.data_ptr(),
                                          twd.data_ptr(), dx.data_ptr(), dgw.data_ptr(), dgb.data_ptr(), dtw.data_ptr(),
                                          dtb.data_ptr(), scr.data_ptr()), "dagl_conv_pair_backward")
        outs.append([t.cpu() for t in (dx, dgw, dgb, dtw, dtb)])
    for got, want, name in zip(outs[0], (xr.grad, gwr.grad, gb.grad, twr.grad, tb.grad), ("d_x", "d_g_w", "d_g_b", "d_th_w", "d_th_b")):
        err = normwise(got.numpy(), want.numpy())
        print(f"conv_pair_backward [{B},64,{H},{W}] {name}: normwise {err:.2e}")
        assert err <= 2e-6, name
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # input gradient alone / parameter gradients alone
    dx2 = torch.empty(B, 64, H, W, device=dev)
    check(lib.dagl_conv_pair_backward(ops._stream(), B, H, W, xd.data_ptr(), d1p.data_ptr(), d2p.data_ptr(), gwd.data_ptr(), twd.data_ptr(),
                                      dx2.data_ptr(), None, None, None, None, None), "dagl_conv_pair_backward")
    assert torch.equal(dx2.cpu(), outs[0][0])
    assert lib.dagl_conv_pair_backward_supported(1, 8, 30) == 0 and lib.dagl_conv_pair_backward_supported(1, 8, 260) == 0


def test_prologue_backward_direct_and_unfold_routes_agree():
    """The module's prologue under autograd: the direct backward of g / theta against the unfold + GEMM + fold route it replaces
    (still taken for widths that are no multiple of 4, and for other channel counts)."""
    from dagl_amd import train_ops as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 36, 40, generator=g).to(dev)
    conv = lambda ci, co, k: torch.nn.Conv2d(ci, co, k, padding=k // 2).to(dev)
    gm, th, thr, bia = conv(64, 16, 3), conv(64, 16, 1), torch.nn.Conv2d(64, 1, 7, stride=4).to(dev), torch.nn.Conv2d(64, 1, 7, stride=4).to(dev)
    res = {}
    for heads in (False, True):
        for force in (False, True):
            T._FORCE_UNFOLD_BACKWARD = force
            try:
                xi = x.clone().requires_grad_(True)
                for m in (gm, th, thr, bia):
                    m.zero_grad()
                outs = T.prologue_convs(xi, gm, th, thr if heads else None, bia if heads else None)
                gen = torch.Generator().manual_seed(9)
                loss = sum((o * torch.randn(o.shape, generator=gen).to(dev)).sum() for o in outs)
                loss.backward()
                res[(heads, force)] = [xi.grad.cpu()] + [p.grad.cpu().clone() for m in (gm, th) for p in m.parameters()]
            finally:
                T._FORCE_UNFOLD_BACKWARD = False
        for a, b in zip(res[(heads, False)], res[(heads, True)]):
            assert normwise(a.numpy(), b.numpy()) <= 2e-6


def test_prologue16_matches_the_fp32_prologue_forward_and_backward():
    """dagl_ce_prologue16 (round 6: g / theta of the differentiable path's forward on the fp16 matrix cores, fp32 map out) against the fp32
    prologue: maps, thr / bias and the gradients through them; an input x 3e3 (beyond the fixed split's range) is served."""
    from dagl_amd import train_ops as T
    from dagl_amd.ce import CE
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    ce = CE(in_channels=64).to(dev)
    for scale in (1.0, 3.0e3):
        x = (torch.randn(2, 64, 40, 44, generator=torch.Generator().manual_seed(7)) * scale).to(dev)
        res = {}
        for fast in (True, False):
            xx = x.clone().requires_grad_(True)
            b1p, b2p, thr, bias = T.prologue_convs(xx, ce.g, ce.theta, ce.thr_conv, ce.bias_conv, fast=fast)
            gsel = torch.Generator().manual_seed(9)
            loss = (b1p * torch.randn(b1p.shape, generator=gsel).to(dev)).sum() + (b2p ** 2).sum() * 1e-3 + thr.sum() + (bias ** 2).sum()
            grads = torch.autograd.grad(loss, [xx, ce.g.weight, ce.theta.weight, ce.thr_conv.weight])
            res[fast] = [t.detach().cpu() for t in (b1p, b2p, thr, bias, *grads)]
        for a, c in zip(res[True], res[False]):
            assert torch.isfinite(a).all()
            assert normwise(a.numpy(), c.numpy()) <= 2e-6, normwise(a.numpy(), c.numpy())
