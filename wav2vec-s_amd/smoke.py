"""One small invocation of the hot path on cuda:0 (used by __graft_entry__.smoke)."""
import torch


def run():
    from . import ops
    torch.manual_seed(0)
    x = torch.randn(300, 256).to(torch.bfloat16)
    w = (torch.randn(384, 256) * 0.1).to(torch.bfloat16)
    b = torch.randn(384).to(torch.bfloat16)
    y = ops.linear_fwd(x.cuda(), w.cuda(), b.cuda())
    ref = x.float() @ w.float().t() + b.float()
    err = float((y.float().cpu() - ref).norm() / ref.norm())
    assert err < 5e-3, err
    print("smoke ok: gemm rel err %.2e" % err)
