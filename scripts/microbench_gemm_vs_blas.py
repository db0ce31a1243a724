"""How fast is the vendor BLAS on the trailing-update shapes (context for gemm.hip's roofline; NOT used by the product)."""
import sys, time, torch
M = N = int(sys.argv[1]) if len(sys.argv) > 1 else 15872
for K in (256, 512, 1024, 4096):
    A = torch.rand((M, K), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((K, N), dtype=torch.float64, device="cuda") - 0.5
    C = torch.rand((M, N), dtype=torch.float64, device="cuda")
    for _ in range(2):
        torch.addmm(C, A, B, beta=1.0, alpha=-1.0, out=C)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); torch.addmm(C, A, B, beta=1.0, alpha=-1.0, out=C); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    print(f"vendor BLAS addmm {M}x{N}x{K}: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:6.2f} TFLOP/s", flush=True)
