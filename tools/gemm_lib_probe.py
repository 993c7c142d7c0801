"""Probe: what do the library GEMMs (hipBLASLt / rocBLAS through torch) reach on the FC1 shape (fp32, M=1024, N=1024, K=7744)
and on the conv GEMM shapes as plain GEMMs (im2col'ed operand already materialised, which the hand kernels never do)?"""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
def bench(M, N, K, reps=30):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    for _ in range(5): torch.mm(a, w.t(), out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): torch.mm(a, w.t(), out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s")
bench(1024, 1024, 7744)
bench(128, 1024, 7744)
bench(1024 * 121, 64, 512)
bench(1024 * 121, 64, 576)
bench(4096, 4096, 4096)
bench(8192, 8192, 8192, reps=5)
