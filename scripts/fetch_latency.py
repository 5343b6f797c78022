"""Dev tool: round trip of Engine.fetch on an idle stream and behind a short kernel (DRLGX_SYNC_SPIN=0/1), against a torch `.cpu()`."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from drl_graph_exploration_amd import default_config
from drl_graph_exploration_amd.engine import Engine
eng = Engine(default_config(40), 16, 0, 0)
t = torch.arange(1024, device=eng.device, dtype=torch.float32)
for name, f in (("fetch", lambda: eng.fetch(t)), ("cpu()", lambda: t.cpu())):
    for busy in (False, True):
        for _ in range(50): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2000):
            if busy: t.add_(1.0)
            f()
        print("%s%s: %.1f us" % (name, " behind a kernel" if busy else " idle", (time.perf_counter() - t0) / 2000 * 1e6))
eng.close()
