"""Second half of __graft_entry__.smoke(): fused SDF kernel vs the CPU oracle and one tiny
training iteration (imports oracle/ - test infrastructure - only from here, never from the product path)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import torch

    import params as P
    from helpers import idr_conf, make_implicit
    from oracle import c_oracle as O
    from hashmodnffbanks_idr_amd import parallel
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss

    # fused SDF forward (every tile size) vs the C oracle
    cfg, hidden, fvs, seed = "tiny", (64,) * 8, 16, 3
    net = make_implicit(cfg, hidden, fvs, seed, 0.5, 0.5, device="cuda:0")
    L, T, b, d = P.CONFIGS[cfg]
    levels, B, _, _ = P.make_embedder_state(seed, cfg, 0.5)
    prm = P.make_sdf_params(seed + 7, 3 + 4 * L, hidden, 1 + fvs, (4,), 0.6, 0.5, 0.1)
    orc = O.SdfOracle(O.Grid(L, T, b, d), np.concatenate(levels, 0), B, prm)
    x = P.make_points(1, 1000)
    ref = orc(x)
    for tile in (4, 8, 16, 64):
        net.sdf_tile_points = tile
        with torch.no_grad():
            out = net(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
        assert np.allclose(out, ref, rtol=1e-5, atol=2e-6), f"fused SDF (tile {tile}) differs from the oracle"

    # one tiny training iteration: device ray tracer + GEMM autograd path + loss + Adam
    torch.manual_seed(0)
    model = IDRNetwork(idr_conf("tiny", hidden=(64,) * 8, fvs=16, rdims=(64,) * 2)).to("cuda:0")
    model.train()
    import bench
    inp, gt = bench.synthetic_batch(7, 128, "cuda:0")
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    out, lo = parallel.train_step(model, IDRLoss(0.1, 100.0, 50.0), opt, inp, gt)
    assert torch.isfinite(lo["loss"]).item() and model.ray_tracer.last_stats["unfinished"] == 0
