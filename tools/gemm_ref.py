"""hipBLASLt reference points for the implicit-GEMM kernels: the dense256 layer's product as a plain GEMM (no gather) and a large
square GEMM, bf16 -> TFLOP/s on this box.  usage (GPU): python tools/gemm_ref.py"""
import torch


def bench(m, n, k, iters=20):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(k, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.mm(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"mm [{m} x {k}] x [{k} x {n}]: {ms:.4f} ms  {2.0 * m * n * k / ms / 1e9:.1f} TF/s")


if __name__ == "__main__":
    bench(192000, 256, 6912)
    bench(192000, 256, 2304)
    bench(8192, 8192, 8192)
    bench(16384, 4096, 4096)
