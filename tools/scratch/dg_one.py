import sys; sys.path.insert(0, "/root/repo")
import torch, tpgan_amd
from tpgan_amd import ops
dev = torch.device("cuda", 0)
hip = ops.backend_for(torch.zeros(1, device=dev))
P, nseg, Cin, Cout, K = 8 * 256 * 32, 6, 128, 256, 32
x_in = torch.randn(P * nseg, Cin, device=dev).bfloat16()
x_out = torch.randn(P * nseg, Cout, device=dev).bfloat16()
W = torch.randn(nseg, Cout, Cin, device=dev) / Cin ** 0.5
ss = torch.rand(nseg, 2, Cin, device=dev)
# forward once to have consistent buffers; then dense dgrad
y, ci_out = hip.mlp_fwd(x_in, ss, 0.01, W, nseg, 1e-5, 0.1, None, None, None, None, torch.ones(nseg, Cout, device=dev), torch.zeros(nseg, Cout, device=dev))
g = torch.randn(P * nseg, Cout, device=dev).bfloat16()
mean = torch.zeros(nseg, Cout, device=dev); rstd = torch.ones(nseg, Cout, device=dev)
gam = torch.ones(nseg, Cout, device=dev); bet = torch.zeros(nseg, Cout, device=dev)
c12 = torch.zeros(nseg, 2, Cout, device=dev)
ci, cb = hip.mlp_consts(mean, rstd, gam, bet, c12, True, True)
cin_ci, _ = hip.mlp_consts(torch.zeros(nseg, Cin, device=dev), torch.ones(nseg, Cin, device=dev), torch.ones(nseg, Cin, device=dev), torch.zeros(nseg, Cin, device=dev), None, True, False)
for _ in range(3):
    hip.mlp_dgrad(y, g, None, 0, cb, x_in, cin_ci, 0.01, W, nseg, True)
torch.cuda.synchronize()
