"""What the vendor library (hipBLASLt / rocBLAS behind torch.matmul) does on the model's plain GEMM shapes -- a yardstick for
csrc/gemm.hip, not a code path of the product."""
import torch, json
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(32000, 3072, 512), (32000, 512, 3072), (32000, 3000, 320), (32000, 640, 512), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ms = t(lambda: torch.matmul(a, b.t()))
    print("nt %dx%dx%d bf16 out: %.4f ms %.1f TF" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
for (K, M, N) in [(32000, 3072, 512), (32000, 1536, 512), (32000, 3000, 320)]:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    ms = t(lambda: torch.matmul(a.t(), b))
    print("tn K%d %dx%d bf16 out: %.4f ms %.1f TF" % (K, M, N, ms, 2.0 * M * N * K / ms / 1e9))
