#!/usr/bin/env python3
"""bench.py - rays/sec fwd+bwd (hash + SDF MLP) and hash-gather HBM GB/s on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.
For N>1 it is launched by torch.distributed.run, one rank per GPU (RCCL).

A "step" = one pass of the hot path over one batch of synthetic uniform-sphere rays
(SURVEY.md section 8d): IDRNetwork.forward (sphere tracing + SDF/rendering MLPs) + IDRLoss +
backward (+ gradient all-reduce when N>1).  Workload = BASELINE.json configs[1]:
MultiResHash L=16 T=2^19 F=2, 2048 rays per GPU, fp32.

Extra objects on the same line:
  roofline     - the hash-gather kernel (BASELINE metric "hash-gather HBM GB/s"): algorithmic
                 bytes (1304 B/point at L=16,F=2) x 2^22 points / HIP-event time per launch
  cpu_baseline - the C oracle (oracle/hm_oracle.c, a port) timed on the host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import params as P  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def gather_bytes_per_point(L, F):
    """12 (x) + L*8*F*4 (corner rows) + (3+2L+L*F)*4 (output row)."""
    return 12 + L * 8 * F * 4 + (3 + 2 * L + L * F) * 4


def build_embedder(cfg, device, seed=0):
    from hashmodnffbanks_idr_amd.model.embeddings.hashGridEmbedding import MultiResHashGridMLP
    L, T, b, d = P.CONFIGS[cfg]
    emb = MultiResHashGridMLP(True, 3, L, 2, T, b, d).to(device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        emb.table.copy_((torch.rand(emb.table.shape, generator=g) * 2e-4 - 1e-4).to(device))
    return emb


def gather_roofline(cfg="C2", log2_n=22, iters=10, warmup=3, device="cuda:0"):
    """HIP-event timing of hm_encode_fwd on torch's current stream (the stream the kernel runs on)."""
    from hashmodnffbanks_idr_amd import ops
    L, T, b, d = P.CONFIGS[cfg]
    emb = build_embedder(cfg, device)
    n = 1 << log2_n
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = (torch.rand((n, 3), generator=g) * 2 - 1).to(device)
    out = None
    for _ in range(warmup):
        out = ops.encode_fwd(emb.desc, x, emb.table.detach(), emb.freq_encoding.B, 0)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        out = ops.encode_fwd(emb.desc, x, emb.table.detach(), emb.freq_encoding.B, 0)
        e.record()
    torch.cuda.synchronize()
    ms = np.asarray([s.elapsed_time(e) for s, e in evs])
    del out
    bpp = gather_bytes_per_point(L, 2)
    avg_ms = float(ms.mean())
    achieved = n * bpp / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel": "encode_fwd_f2_kernel", "points": n, "bytes_per_point": bpp,
            "avg_launch_ms": round(avg_ms, 4), "min_launch_ms": round(float(ms.min()), 4),
            "workload": f"{cfg}: L={L} T=2^{T} F=2, N=2^{log2_n} points U([-1,1]^3), table "
                        f"{emb.table.numel() * 4 / 2**20:.1f} MiB (cache-resident below 256 MiB)"}


def cpu_baseline_gather(cfg="C2", n=1 << 18):
    """The C oracle (port) on the host cores, same synthetic inputs, bounded sample."""
    from oracle import c_oracle as O
    L, T, b, d = P.CONFIGS[cfg]
    grid = O.Grid(L, T, b, d)
    rs = np.random.RandomState(0)
    table = rs.uniform(-1e-4, 1e-4, (grid.total_rows, 2)).astype(np.float32)
    B = P.make_fourier_B(1, L, P.fourier_sigma(b, d))
    x = P.make_points(1234, n)
    O.encode_fwd(grid, x[:1024], table, B, 0)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 5.0:
        O.encode_fwd(grid, x, table, B, 0)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"points_per_s": n / dt, "gbs": n * gather_bytes_per_point(L, 2) / dt / 1e9,
            "cores": O.lib().hmo_num_threads(), "sample_points": n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gather-only", action="store_true", help="only the hash-gather roofline section")
    ap.add_argument("--gather-log2n", type=int, default=22)
    ap.add_argument("--gather-cfg", default="C2")
    args = ap.parse_args()

    if args.gather_only:
        r = gather_roofline(args.gather_cfg, args.gather_log2n)
        c = cpu_baseline_gather(args.gather_cfg)
        print(json.dumps({"roofline": r, "cpu_gather": c}))
        return
    raise SystemExit("full ray step not wired yet")


if __name__ == "__main__":
    main()
