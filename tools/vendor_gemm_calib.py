"""GPU box: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, plain bf16 GEMM, NO epilogue) takes on the transformer GEMM
shapes of the 0.25-degree model, beside the engine's own per-class launch times (which include LayerNorm fold / bias / GELU /
residual / row statistics).   python tools/vendor_gemm_calib.py"""
import torch

shapes = [("s2 to_qkv", 20000, 1536, 512), ("s2 FeedForward 1", 20000, 2048, 512), ("s2 to_out", 20000, 512, 512),
          ("s2 FeedForward 2", 20000, 512, 2048), ("s3 to_qkv", 5000, 3072, 1024), ("s3 FeedForward 1", 5000, 4096, 1024),
          ("s3 to_out", 5000, 1024, 1024), ("s3 FeedForward 2", 5000, 1024, 4096),
          ("s0 FeedForward 1", 320000, 512, 128), ("s1 FeedForward 1", 80000, 1024, 256)]
print(f"torch {torch.__version__}, preferred BLAS backend: {getattr(torch.backends.cuda, 'preferred_blas_library', lambda: '?')()}")
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, w.t(), out=out)
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.matmul(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    print(f"{name:<18s} M={M:6d} N={N:5d} K={K:5d}  {best:7.1f} us  {2.0 * M * N * K / best * 1e-6:6.0f} TFLOP/s")
