# what the vendor library (hipBLASLt through torch.matmul) reaches on this pipeline's GEMM shapes: a yardstick for
# gemm_nt_glds_kernel, not a product path
import torch, time
shapes = [("dec qkv", 12992, 4096, 1024), ("dec o", 12992, 1024, 2048), ("dec gate/up", 12992, 6144, 1024),
          ("dec down", 12992, 1024, 3072), ("enc qkv", 12480, 2688, 896), ("enc o", 12480, 896, 896),
          ("enc fc1", 12480, 3584, 896), ("enc fc2", 12480, 896, 3584), ("conv2 (im2col)", 768000, 480, 4320),
          ("conv_out", 12480, 896, 7680)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): c = a @ w.T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): c = a @ w.T
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:16s} {M}x{N}x{K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.0f} TFLOP/s", flush=True)
