import torch, sys
BF = torch.bfloat16
def t_us(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
R = 6544
for name, N, K in (("fc1", 3072, 768), ("qkv", 2304, 768), ("fc2", 768, 3072), ("out", 768, 768), ("conv1", 512, 1536)):
    M = R if name != "conv1" else 8 * 17500
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.03).to(BF); b = torch.randn(N, device="cuda").to(BF)
    t = t_us(lambda: torch.nn.functional.linear(x, w, b))
    t2 = t_us(lambda: torch.matmul(x, w.t()))
    print("%-6s M=%6d N=%4d K=%4d  torch linear+bias %6.1f us (%4.0f TF/s)   matmul %6.1f us (%4.0f TF/s)" % (name, M, N, K, t, 2.0*M*N*K/t/1e6, t2, 2.0*M*N*K/t2/1e6), flush=True)
    dy = torch.randn(M, N, device="cuda").to(BF)
    t3 = t_us(lambda: torch.matmul(dy.t(), x))
    print("       wgrad  %6.1f us (%4.0f TF/s)" % (t3, 2.0*M*N*K/t3/1e6), flush=True)
