"""Development aid: f32 GEMM rates of the PPO update shapes (K = 617 vs padded 640)."""
import torch, time
dev = 'cuda'
def rate(M, K, N, iters=20):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    for _ in range(3): (a @ w.t())
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): (a @ w.t())
    torch.cuda.synchronize(); dt = (time.time() - t0) / iters
    return 2 * M * K * N / dt / 1e12, dt * 1e3
for (M, K, N) in [(53248, 617, 2048), (53248, 640, 2048), (53248, 2048, 1024), (53248, 1024, 512), (53248, 512, 32), (4096, 617, 2048), (4096, 640, 2048)]:
    print((M, K, N), 'fwd TF/s %.1f ms %.3f' % rate(M, K, N))
# dW shape: [N,M]@[M,K]
def rate_dw(M, K, N, iters=20):
    g = torch.randn(M, N, device=dev); a = torch.randn(M, K, device=dev)
    for _ in range(3): (g.t() @ a)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): (g.t() @ a)
    torch.cuda.synchronize(); dt = (time.time() - t0) / iters
    return 2 * M * K * N / dt / 1e12, dt * 1e3
for (M, K, N) in [(53248, 617, 2048), (53248, 640, 2048), (53248, 2048, 1024)]:
    print((M, K, N), 'dW  TF/s %.1f ms %.3f' % rate_dw(M, K, N))
