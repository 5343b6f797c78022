import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import torch
from oracle import gcn_ref
from test_gpu_gcn import random_batch, make_params, rel_err
from drl_graph_exploration_amd.networks import gcn_trunk
dev = torch.device("cuda", 0)
for ng in (30, 64, 200):
    x, ei, ea, batch = random_batch(ng, 123 + ng, dev)
    P = make_params(dev, 1)
    N = x.shape[0]
    out = gcn_trunk(x, ei, ea, *[P[k] for k in ("conv1.weight","conv1.bias","conv2.weight","conv2.bias","fully_con1.weight","fully_con1.bias")], None)
    wgt = torch.randn(N, 1, device=dev)
    (out * wgt).sum().backward()
    r32 = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    ref32 = gcn_ref.gcn_forward(r32, x, ei, ea); (ref32 * wgt).sum().backward()
    r64 = {k: v.detach().double().clone().requires_grad_(True) for k, v in P.items()}
    ref64 = gcn_ref.gcn_forward(r64, x.double(), ei, ea.double()); (ref64 * wgt.double()).sum().backward()
    print("graphs", ng, "N", N, "fwd: mine-vs-64 %.2e torch32-vs-64 %.2e" % (rel_err(out.detach().double(), ref64.detach()), rel_err(ref32.detach().double(), ref64.detach())))
    for k in P:
        print("   %-18s mine-vs-64 %.2e   torch32-vs-64 %.2e" % (k, rel_err(P[k].grad.double(), r64[k].grad), rel_err(r32[k].grad.double(), r64[k].grad)))
