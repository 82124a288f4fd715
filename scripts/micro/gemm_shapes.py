"""fp32 GEMM times of the time-MLPs' shapes (7 600 samples x 256 hidden units: 10 envs x 760 steps) under the BLAS back ends torch offers
on this stack.  python scripts/micro/gemm_shapes.py"""
import sys, time, torch
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7600
shapes = [("fwd  x[N,256] W^T[256,256] +b", lambda: (torch.randn(N, 256, device=dev), torch.randn(256, 256, device=dev), torch.randn(256, device=dev)), lambda x, w, b: torch.addmm(b, x, w.t())),
          ("fwd  x[N,269] W^T[269,256] +b", lambda: (torch.randn(N, 269, device=dev), torch.randn(256, 269, device=dev), torch.randn(256, device=dev)), lambda x, w, b: torch.addmm(b, x, w.t())),
          ("bwd  g[N,256] W[256,256]", lambda: (torch.randn(N, 256, device=dev), torch.randn(256, 256, device=dev)), lambda g, w: g @ w),
          ("bwd  g^T[256,N] x[N,256]", lambda: (torch.randn(N, 256, device=dev), torch.randn(N, 256, device=dev)), lambda g, x: g.t() @ x),
          ("bwd  ones[1,N] g[N,256]", lambda: (torch.ones(1, N, device=dev), torch.randn(N, 256, device=dev)), lambda o, g: o @ g),
          ("head x[N,256] W^T[256,12]", lambda: (torch.randn(N, 256, device=dev), torch.randn(12, 256, device=dev), torch.randn(12, device=dev)), lambda x, w, b: torch.addmm(b, x, w.t()))]
for lib in ("default", "hipblas", "hipblaslt"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib, "unavailable:", e); continue
    for name, mk, f in shapes:
        a = mk()
        for _ in range(5): f(*a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f(*a)
        e1.record(); torch.cuda.synchronize()
        print("%-10s %-34s %7.1f us" % (lib, name, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
