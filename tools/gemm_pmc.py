import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grates_amd as ga
A = torch.rand((8192, 8192), dtype=torch.float64, device='cuda') - 0.5
B = torch.rand((8192, 8192), dtype=torch.float64, device='cuda') - 0.5
for _ in range(3):
    ga.engine.dgemm(A, B)
torch.cuda.synchronize()
