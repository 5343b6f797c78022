"""Dev tool: per-parameter gradient errors of the HIP GCN trunk and of the plain fp32 torch evaluation against the float64 ground truth
(tests/test_gpu_gcn.py's comparison, printed instead of asserted): python scripts/gcn_grad_errors.py N out_dim [seeds]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_gcn as T
from drl_graph_exploration_amd.networks import gcn_trunk
gcn_ref = T.gcn_ref
dev = torch.device("cuda", 0)
N, od = int(sys.argv[1]), int(sys.argv[2])
for seed in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    if os.environ.get("AS_TEST"):  # the parametrised test's own instance for (N, out_dim)
        x, ei, ea = T.batch_of_about(N, 77 + N, dev)
        mask = (torch.rand(N, 1000, device=dev, generator=torch.Generator(device=dev).manual_seed(11 + N)) >= 0.5).float() * 2.0
        wgt = torch.randn(N, od, device=dev, generator=torch.Generator(device=dev).manual_seed(3 + N + seed))
    else:
        x, ei, ea = T.batch_of_about(N, 77 + seed, dev)
        g = torch.Generator(device=dev).manual_seed(seed)
        mask = (torch.rand(N, 1000, device=dev, generator=g) >= 0.5).float() * 2.0
        wgt = torch.randn(N, od, device=dev, generator=g)
    P = T.make_params(dev, od)
    out = gcn_trunk(x, ei, ea, P["conv1.weight"], P["conv1.bias"], P["conv2.weight"], P["conv2.bias"], P["fully_con1.weight"], P["fully_con1.bias"], mask)
    (out * wgt).sum().backward()
    rp = {k: v.detach().double().clone().requires_grad_(True) for k, v in P.items()}
    (gcn_ref.gcn_forward(rp, x.double(), ei, ea.double(), mask.double()) * wgt.double()).sum().backward()
    p32 = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    (gcn_ref.gcn_forward(p32, x, ei, ea, mask) * wgt).sum().backward()
    print("seed", seed, " ".join("%s %.1e/%.1e" % (k.replace("fully_con1", "f").replace("weight", "w").replace("bias", "b"),
          float((P[k].grad.double() - rp[k].grad).norm() / rp[k].grad.norm()), float((p32[k].grad.double() - rp[k].grad).norm() / rp[k].grad.norm())) for k in P))
