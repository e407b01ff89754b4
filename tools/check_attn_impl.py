import os, sys, importlib, numpy as np, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/tests') else os.environ.get('GRAFT_REPO_ROOT', '.'))
ltx = importlib.import_module('ltx-video-swift-mlx_amd')
ctx = ltx.Context(0)
def run(impl, T, S, H, seed=0):
    ltx.set_option('attn_impl', int(impl))
    g = torch.Generator(device='cuda').manual_seed(seed)
    Q = torch.randn(1, T, H*128, device='cuda', generator=g).to(torch.bfloat16)
    K = torch.randn(1, S, H*128, device='cuda', generator=g).to(torch.bfloat16)
    Vt = torch.randn(1, H*128, S, device='cuda', generator=g).to(torch.bfloat16)
    O = torch.empty(1, T, H*128, device='cuda', dtype=torch.bfloat16)
    ctx.op_attention(Q, K, Vt, None, H, O)
    torch.cuda.synchronize()
    return O.float()
impl = sys.argv[1] if len(sys.argv) > 1 else '3'
for (T, S, H) in [(192, 256, 1), (384, 512, 2), (1536, 1536, 4), (1536, 1024, 3)]:
    a = run('1', T, S, H); b = run(impl, T, S, H)
    err = (a - b).abs().max().item(); rel = ((a - b).norm() / a.norm()).item()
    print(f"T={T} S={S} H={H}: max abs diff {err:.4e} rel-L2 {rel:.3e}", flush=True)
    assert rel < 5e-3, "MISMATCH"
print("ok")
