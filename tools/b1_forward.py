"""One-clip forwards for a profiler run: `rocprofv3 --kernel-trace --stats -- python3 tools/b1_forward.py [iters] [F T] [batch]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from audiodenoiser_amd.weights import make_state_dict  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
f, t = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (513, 256)
b = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0)
net = bench.make_net(make_state_dict(1234), dev, os.environ.get("B1_DTYPE", "f32"))
x = torch.rand((b, 1, f, t), device=dev) * 4
with torch.no_grad():
    for _ in range(iters):
        y = net(x)
torch.cuda.synchronize()
print("ok", float(y.abs().max()))
