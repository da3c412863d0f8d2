"""How long does one dependent kernel boundary cost on this stack? (graph replay of N trivial torch kernels)"""
import time, torch
x = torch.zeros(1, device="cuda")
y = torch.zeros(1 << 22, device="cuda")   # 16 MB
for (name, fn, n) in (("1-element add_", lambda: x.add_(1.0), 500), ("16 MB add_", lambda: y.add_(1.0), 200)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(10): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): g.replay()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: graph {dt / reps / n * 1e6:.2f} us per kernel")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n * 5): fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: eager {dt / (n * 5) * 1e6:.2f} us per kernel")

# the same question through libcae_hip: n no-op kernels captured on the engine's stream
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cae_tools_amd.engine import HipEngine
from bench import build_model, FC, LATENT
spec, enc, dec = build_model(0)
eng = HipEngine(spec, FC, LATENT, max_batch=64)
out = C.c_double()
for n in (1, 20, 500):
    eng.lib.cae_debug_launch_floor(eng.handle, n, C.byref(out))
    print(f"libcae_hip no-op kernel x{n} in a graph: {out.value:.2f} us per kernel")
