import sys; sys.path.insert(0, "/root/repo")
import torch, tpgan_amd
from tpgan_amd import ops
dev = torch.device("cuda", 0)
hip = ops.backend_for(torch.zeros(1, device=dev))
P, nseg, Cin, Cout = 8 * 256 * 32, 6, 128, 256
x_in = torch.randn(P * nseg, Cin, device=dev).bfloat16()
W = torch.randn(nseg, Cout, Cin, device=dev) / Cin ** 0.5
ss = torch.rand(nseg, 2, Cin, device=dev)
for _ in range(3):
    hip.mlp_fwd(x_in, ss, 0.01, W, nseg, 1e-5, 0.1, None, None, None, None, torch.ones(nseg, Cout, device=dev), torch.zeros(nseg, Cout, device=dev))
torch.cuda.synchronize()
