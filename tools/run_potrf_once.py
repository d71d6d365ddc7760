import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
m = 8448
g = torch.Generator().manual_seed(0)
B = torch.rand((m, 512), generator=g, dtype=torch.float64).cuda()
S = B @ B.t() + torch.eye(m, dtype=torch.float64, device="cuda") * 50.0
for _ in range(2):
    L = S.clone(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.potrf_inv(L); e1.record(); torch.cuda.synchronize()
    print("potrf_inv m=%d: %.2f ms" % (m, e0.elapsed_time(e1)))
