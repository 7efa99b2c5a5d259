"""Which Python lines of the static step issue the ATen launches that remain in it?  One eager static iteration (C2,
bench batch) under torch.profiler (CPU activity, with_stack): leaf aten ops that launch a kernel, grouped by the innermost
frame of this package (forward) or by the autograd node (backward).  Run on the GPU box: python scripts/glue_sources.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hashmodnffbanks_idr_amd.model.loss import IDRLoss  # noqa: E402
from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep  # noqa: E402
from hashmodnffbanks_idr_amd.training.optim import ClipAdam  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda", 0)
model = bench._build(cfg, dev, 0.0)
model.ray_tracer.sampler_head = 0
rays = bench.NFFB_CONFIGS[cfg][5] if cfg in getattr(bench, "NFFB_CONFIGS", {}) else bench.RAYS_PER_GPU
inp, gt = bench.synthetic_batch(1234, rays, dev)
opt, lfn = ClipAdam(model.parameters(), lr=0.0, max_norm=1.0), IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
st = GraphedTrainStep(model, lfn, opt, None, warmup=10 ** 9)      # eager static iterations only
for _ in range(3):
    st.step(inp, gt)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

NOLAUNCH = ("view", "reshape", "slice", "select", "expand", "as_strided", "detach", "t.default", "transpose", "permute",
            "unsqueeze", "squeeze", "empty", "alias", "_unsafe_view", "lift_fresh", "split", "unbind", "is_", "size", "stride",
            "numel", "_local_scalar", "_to_copy", "new_empty", "empty_like", "empty_strided", "narrow", "unfold", "result_type",
            "set_", "resize_", "record_stream", "is_pinned", "_has_compatible")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sites = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in NOLAUNCH):
            node = torch._C._current_autograd_node()
            site = None
            for fr in reversed(traceback.extract_stack()):
                if root in fr.filename and "scripts/" not in fr.filename:
                    site = f"{fr.filename.replace(root + '/', '')}:{fr.lineno}"
                    break
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            sites[(name, type(node).__name__ if node is not None else "-", site or "?", str(shapes))] += 1
        return func(*args, **(kwargs or {}))


with Rec():
    st.step(inp, gt)
torch.cuda.synchronize()
print(f"{sum(sites.values())} dispatched ops that may launch")
for (name, node, site, shapes), c in sorted(sites.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print(f"{c:3d}  {name:32s} {node:28s} {site:70s} {shapes}")
