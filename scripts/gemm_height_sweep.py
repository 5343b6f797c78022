"""Dev tool: the layer-2 forward product (M x 1000 x 1000, bias + ReLU + mask epilogue) at a list of batch sizes, the tall tile's height
pinned by DRLGX_GEMM_WIDE=6..10 (one process per height; run under rocprofv3 --kernel-trace and read the k_gemm_wide durations in launch
order with scripts/rocpd_kernel_summary.py-style queries): python scripts/gemm_height_sweep.py M0 M1 STEP"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from test_gpu_gcn import batch_of_about, make_params
from drl_graph_exploration_amd.networks import gcn_trunk
dev = torch.device("cuda", 0)
P = make_params(dev, 1)
for n in range(int(sys.argv[1]), int(sys.argv[2]) + 1, int(sys.argv[3])):
    x, ei, ea = batch_of_about(n, 5, dev)
    mask = (torch.rand(n, 1000, device=dev) >= 0.5).float() * 2.0
    with torch.no_grad():
        for it in range(3):
            gcn_trunk(x, ei, ea, P["conv1.weight"], P["conv1.bias"], P["conv2.weight"], P["conv2.bias"], P["fully_con1.weight"], P["fully_con1.bias"], mask)
torch.cuda.synchronize()
