"""GPU tier: sc_pointwise_block_forward (k_pblock_fwd) on the device -- the whole pointwise side of a default FNO
block's forward in one pass (fno_block.py:392-412: linear skip, add, GELU, ChannelMLP, soft-gating skip, GELU) against
float64 torch, and its s / y outputs against the passes it replaces (bit for bit)."""
import pytest
import torch
import torch.nn.functional as F

from neuraloperator_amd import _lib
from engine_runner import rel_l2

pytestmark = pytest.mark.gpu
TOL = 3e-6


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("GPU tier: no GPU visible")
    return _lib.get_lib()


@pytest.mark.parametrize("c,ch,B,S", [(64, 32, 32, 256 * 256), (64, 64, 3, 4096), (32, 32, 5, 96), (64, 32, 2, 32)], ids=str)
@pytest.mark.parametrize("act", [1, 0])
def test_block_pass_on_device(lib, c, ch, B, S, act):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(c * 7 + ch + act)
    mk = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x, conv = mk(B, c, S), mk(B, c, S)
    ws, w1, w2 = mk(c, c) / c ** 0.5, mk(ch, c) / c ** 0.5, mk(c, ch) / ch ** 0.5
    bs, b1, b2, gt = mk(c), mk(ch), mk(c), mk(c)
    y, pre, out = (torch.full((B, c, S), float("nan"), device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    lib.pointwise_block_forward(B, c, ch, S, act, p(conv), p(x), p(ws), p(bs), p(w1), p(b1), p(w2), p(b2), p(gt), p(y),
                                p(pre) if act else 0, p(out), st)
    # the passes it replaces: linear skip, add (+ GELU), MLP pass
    skip, out3 = torch.empty_like(x), torch.empty_like(x)
    lib.pointwise_linear_forward(B, c, c, S, p(x), p(ws), p(bs), p(skip), st)
    s3 = conv + skip
    torch.cuda.synchronize()
    assert torch.equal(pre if act else y, s3)
    lib.pointwise_mlp_forward(B, c, ch, c, S, act, p(y), p(w1), p(b1), p(w2), p(b2), p(x), p(gt), p(out3), st)
    torch.cuda.synchronize()
    assert rel_l2(out.cpu().numpy(), out3.cpu().numpy()) < TOL
    # float64 (a slice of the large case)
    nb = min(B, 2)
    xd, cd = x[:nb].double(), conv[:nb].double()
    sd = cd + torch.einsum("oc,bcs->bos", ws.double(), xd) + bs.double()[None, :, None]
    yd = F.gelu(sd) if act else sd
    hd = F.gelu(torch.einsum("hc,bcs->bhs", w1.double(), yd) + b1.double()[None, :, None])
    zd = torch.einsum("oh,bhs->bos", w2.double(), hd) + b2.double()[None, :, None] + gt.double()[None, :, None] * xd
    od = F.gelu(zd) if act else zd
    assert rel_l2(y[:nb].cpu().numpy(), yd.cpu().numpy()) < TOL
    assert rel_l2(out[:nb].cpu().numpy(), od.cpu().numpy()) < TOL
