"""InfoNCE-only driver for rocprofv3 --pmc passes: B=1024, d=64 forward + backward, 5 rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops
B, d = 1024, 64
for _ in range(5):
    z1 = torch.randn(B, d, device="cuda", requires_grad=True)
    z2 = torch.randn(B, d, device="cuda", requires_grad=True)
    ops.infonce(z1, z2, 0.5).backward()
torch.cuda.synchronize()
