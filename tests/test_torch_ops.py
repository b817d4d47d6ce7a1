"""torch.ops.mmgt_hip.* (mmgt_amd/torch_ops.py): dispatcher registration over the C ABI."""
import pytest
import torch
import torch.nn.functional as F

import mmgt_amd.torch_ops as T
from mmgt_amd.synthetic import hash_uniform


def test_ops_are_registered_with_schemas_and_have_no_cpu_kernel():
    for name in T.OPS:
        op = getattr(torch.ops.mmgt_hip, name)
        assert "mmgt_hip::" + name in str(op.default._schema)
    with pytest.raises(NotImplementedError):
        torch.ops.mmgt_hip.gemm(torch.zeros(4, 64), torch.zeros(8, 64), None, None, 0)


def test_fake_kernels_give_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(6, 8, 8, 64, device="cuda", dtype=torch.bfloat16)
        w = torch.empty(128, 9 * 64, device="cuda", dtype=torch.bfloat16)
        assert torch.ops.mmgt_hip.conv3x3_nhwc(x, w, None, None, 2, False).shape == (6, 4, 4, 128)
        assert torch.ops.mmgt_hip.conv3x3_nhwc(x, w, None, None, 1, True).shape == (6, 16, 16, 128)
        a = torch.empty(10, 64, device="cuda")
        assert torch.ops.mmgt_hip.gemm(a, torch.empty(128, 64, device="cuda"), None, None, 1).shape == (10, 64)


@pytest.mark.gpu
def test_ops_match_torch_math_fp32():
    dev = "cuda:0"
    r = lambda n, s, sc=1.0: hash_uniform(n, s, sc).to(dev)
    a, w, b, res = r("a", (200, 320)), r("w", (640, 320), 0.05), r("b", (640,)), r("res", (200, 640))
    torch.testing.assert_close(torch.ops.mmgt_hip.gemm(a, w, b, res, 0), F.linear(a, w, b) + res, rtol=1e-3, atol=1e-4)
    q, k, v = r("q", (3, 50, 320)), r("k", (3, 70, 320)), r("v", (3, 70, 320))
    sp = lambda t: t.view(3, -1, 8, 40).transpose(1, 2)
    want = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(3, 50, 320)
    torch.testing.assert_close(torch.ops.mmgt_hip.attention(q, k, v, 8, 40 ** -0.5), want, rtol=1e-3, atol=1e-4)
    x = r("x", (2, 64, 320), 2.0)
    g, be = r("g", (320,), 0.2) + 1, r("be", (320,), 0.2)
    want = F.silu(F.group_norm(x.permute(0, 2, 1), 32, g, be, 1e-5)).permute(0, 2, 1)
    torch.testing.assert_close(torch.ops.mmgt_hip.groupnorm_silu(x, g, be, 32, 1e-5, True), want, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(torch.ops.mmgt_hip.layernorm(x.view(128, 320), g, be, 1e-5),
                               F.layer_norm(x.view(128, 320), (320,), g, be, 1e-5), rtol=1e-3, atol=1e-4)
    from mmgt_amd.packing import pack_conv3x3
    xi, wc = r("xi", (2, 8, 8, 64)), r("wc", (128, 64, 3, 3), 0.05)
    want = F.conv2d(xi.permute(0, 3, 1, 2), wc, None, stride=2, padding=1).permute(0, 2, 3, 1)
    got = torch.ops.mmgt_hip.conv3x3_nhwc(xi, pack_conv3x3(wc).to(dev), None, None, 2, False)
    torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-4)
