import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import recursivefactorization.jl_amd as rf
n = 16384
A = torch.rand((n, n), dtype=torch.float64, device="cuda")
F = rf.lu_(A, None, True, check=False)
B0 = torch.rand((n, 64), dtype=torch.float64, device="cuda")
for _ in range(3):
    X = B0.clone(); rf.ldiv_(F, X); torch.cuda.synchronize()
