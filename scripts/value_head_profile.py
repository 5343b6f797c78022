"""Dev tool: forward + backward of the GCN trunk with the critic's 100-wide read-out on a 12.8 k-node batch (an A2C update chunk of 256
graphs), a few times - to be run under `rocprofv3 --kernel-trace` and summarised by scripts/rocpd_kernel_summary.py."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_gcn import batch_of_about, make_params
from drl_graph_exploration_amd.networks import gcn_trunk
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
x, ei, ea = batch_of_about(n, 5, dev)
P = make_params(dev, 100)
mask = (torch.rand(x.shape[0], 1000, device=dev) >= 0.5).float() * 2.0
wgt = torch.randn(x.shape[0], 100, device=dev)
for it in range(6):
    out = gcn_trunk(x, ei, ea, P["conv1.weight"], P["conv1.bias"], P["conv2.weight"], P["conv2.bias"], P["fully_con1.weight"], P["fully_con1.bias"], mask)
    (out * wgt).sum().backward()
torch.cuda.synchronize()
