#!/usr/bin/env python3
"""bench.py - rays/sec fwd+bwd (hash + SDF MLP) and hash-gather HBM GB/s on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.
For N>1 it is launched by torch.distributed.run, one rank per GPU (RCCL over xGMI); started WITHOUT that launcher
(`python bench.py --gpus N`, no WORLD_SIZE in the environment) it starts the N ranks itself as a child
`python -m torch.distributed.run ...` before anything touches the GPU and relays rank 0's line.

A "step" = one training iteration of the hot path on one batch of synthetic uniform-sphere rays
(SURVEY.md section 8d), in the reference runner's order (idr_train.py:294-308):
IDRNetwork.forward (sphere tracing with the fused SDF kernel, grad-enabled SDF / rendering MLPs,
gradient() with create_graph) + IDRLoss + backward + [gradient all-reduce] + clip_grad_norm_ + Adam.
Workload = BASELINE.json configs[1]: MultiResHash L=16 T=2^19 F=2, 2048 rays PER GPU (weak
scaling), fp32, geometric-init weights seed 0, object_mask all true, rgb_gt = 0 - the SAME table for every N, so the
1 -> 8 curve compares like with like; BASELINE configs[3] (T=2^22; 16 384 rays over 8 GPUs = the same 2048 rays per
GPU) is `--cfg C4` and is reported beside the headline as `config4_leg` at every N.

Two legs are timed, each over exactly K steps:
  value / ms_per_step   the section-8(d) workload: the weights STAY at the geometric initialisation (the whole
                        iteration runs, optimizer included, with lr = 0), so every timed step traces the same
                        surface: ~123 SDF evaluations per ray;
  train_leg             the same iteration with lr = 1e-4 (Adam moves the surface while the clock runs, the per-ray
                        work drifts with the step count - reported, never the headline).
config.sdf_evals_per_step holds the measured SDF evaluations of the timed steps (mean / min / max).

Extra objects on the same line:
  roofline      hash-gather kernel (BASELINE metric "hash-gather HBM GB/s"): algorithmic bytes
                (1304 B/point at L=16,F=2) x 2^22 points / HIP-event time per launch
  roofline_c4   the same kernel over the configs[3] table (T=2^22: 223.5 MiB, larger than the 32 MiB of L2)
  roofline_bwd  table-gradient scatter: 2188 B/point
  roofline_mlp  fused SDF forward kernel (the kernel that dominates the step): 3.93 MFLOP/point
                against the 157.3 TFLOP/s fp32 MFMA peak
  cpu_baseline  oracle/torch_ref.py (a port: the reference's op sequence on torch-CPU) timed on
                the host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import params as P  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
SDF_MAC_PER_POINT = 1966592  # SURVEY.md 8a row A9 (independent of the embedding width)
RAYS_PER_GPU = 2048
LAZY_SAMPLER_HEAD = int(os.environ.get("HM_LAZY_HEAD", "16"))   # RayTracing.sampler_head of the product (model/ray_tracing.py); legs other than "lazy" use 0
CFG = "C2"          # BASELINE.json configs[1] at EVERY --gpus N (one table for the whole scaling curve); configs[3] = --cfg C4


class Conf(dict):
    def _g(self, k):
        d = self
        for p in k.split("."):
            d = d[p]
        return d

    def get_int(self, k):
        return int(self._g(k))

    def get_float(self, k):
        return float(self._g(k))

    def get_config(self, k):
        v = self.get(k)
        return Conf(v) if v is not None else None


# measurement-only configurations (not golden-pinned): C2's level structure with every table L2-resident
EXTRA_CONFIGS = {"C2_l2": (16, 15, 16, 512)}
# BASELINE.json configs[2] / configs[4]: the filter-bank embedders with the shipped embedding_network block
# (confs/embedder_conf_var/{FFB,StyleModNFFB}/dtu_fixed_cameras.conf: L=6, T=5, base 16, desired 512)
NFFB_CONFIGS = {"C3": ("FFB", 6, 5, 16, 512, 4096), "C5": ("StyleModNFFB", 6, 5, 16, 512, 2048)}


def idr_conf(cfg):
    """confs/embedder_conf_var/MultiResHashPointsAndViewDirs/dtu_fixed_cameras.conf with the
    embedding_network block of BASELINE.json's config."""
    embed_type = "HashGrid"
    if cfg in NFFB_CONFIGS:
        embed_type, L, T, b, d, _ = NFFB_CONFIGS[cfg]
    else:
        L, T, b, d = P.CONFIGS[cfg] if cfg in P.CONFIGS else EXTRA_CONFIGS[cfg]
    return Conf(
        feature_vector_size=256,
        implicit_network=dict(d_in=3, d_out=1, dims=[512] * 8, geometric_init=True, bias=0.6, skip_in=[4],
                              weight_norm=True, multires=L),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, viewdirs_embed_type="HashGrid", dims=[512] * 4,
                               weight_norm=True, multires_view=4),
        ray_tracer=dict(object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=3,
                        sphere_tracing_iters=10, n_steps=100, n_secant_steps=8),
        embedding_network=dict(embed_type=embed_type, log2_max_hash_size=T, max_points_per_entry=2,
                               base_resolution=b, desired_resolution=d, bound=1.0),
    )


def synthetic_batch(seed, n_rays, device):
    """Uniform-sphere rays through an identity-K pinhole (uv reproduce the directions exactly enough)."""
    cam, dirs = P.make_rays(seed, n_rays)
    z = -cam[0] / np.linalg.norm(cam[0])
    up = np.array([0.0, 1.0, 0.0])
    xax = np.cross(up, z)
    xax /= np.linalg.norm(xax)
    yax = np.cross(z, xax)
    R = np.stack([xax, yax, z], 1)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R
    pose[:3, 3] = cam[0]
    dc = dirs[0].astype(np.float64) @ R
    uv = (dc[:, :2] / dc[:, 2:3]).astype(np.float32).reshape(1, n_rays, 2)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    inp = dict(intrinsics=T(np.eye(4, dtype=np.float32).reshape(1, 4, 4)), uv=T(uv), pose=T(pose.reshape(1, 4, 4)),
               object_mask=torch.ones((1, n_rays), dtype=torch.bool, device=device))
    gt = dict(rgb=torch.zeros((1, n_rays, 3), dtype=torch.float32, device=device))
    return inp, gt


def gather_bytes_per_point(L, F):
    """12 (x) + L*8*F*4 (corner rows) + (3+2L+L*F)*4 (output row)."""
    return 12 + L * 8 * F * 4 + (3 + 2 * L + L * F) * 4


def gather_roofline(emb, log2_n=22, iters=10, warmup=3):
    """HIP-event timing of hm_encode_fwd on torch's current stream (the stream the kernel runs on)."""
    from hashmodnffbanks_idr_amd import ops
    dev = emb.table.device
    n = 1 << log2_n
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = (torch.rand((n, 3), generator=g) * 2 - 1).to(dev)
    table, B = emb.table.detach(), emb.freq_encoding.B
    for _ in range(warmup):
        out = ops.encode_fwd(emb.desc, x, table, B, 0)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        out = ops.encode_fwd(emb.desc, x, table, B, 0)
        e.record()
    torch.cuda.synchronize()
    del out
    ms = np.asarray([s.elapsed_time(e) for s, e in evs])
    bpp = gather_bytes_per_point(emb.n_levels, emb.n_features)
    avg_ms = float(ms.mean())
    achieved = n * bpp / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    # PMC counters cannot be read from inside this process: the number comes from the separate
    # rocprofv3 --pmc passes of `bench.py --only gather [--cfg C4]` recorded under profiles/ (same kernel, same launch)
    tag = {5217937: "C2", 29295887: "C4"}.get(int(emb.table.shape[0]))
    for rnd in ("r03", "r02", "r01"):
        name = f"{rnd}_gather_pmc.json" if tag == "C2" else f"{rnd}_gather_{tag}_pmc.json"
        pmc = os.path.join(ROOT, "profiles", name)
        if tag and os.path.exists(pmc) and log2_n == 22 and emb.n_levels == 16:
            rec = json.load(open(pmc))
            traffic, traffic_src = round(rec["traffic_bytes_per_launch_corrected"]), "profiles/" + name
            break
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": n * bpp,
            "kernel": "encode_fwd_f2_zorder_kernel (+ 3 bucketing passes, inside the timed launch)"
            if (emb.table.numel() * 4 > (8 << 20) and n >= 512 * 256) else "encode_fwd_f2_kernel",
            "units_per_launch": n, "bytes_per_unit": bpp, "avg_launch_ms": round(avg_ms, 4),
            "min_launch_ms": round(float(ms.min()), 4),
            "note": f"table {emb.table.numel() * 4 / 2**20:.1f} MiB (< 256 MiB Infinity Cache); big launches walk the points "
                    "in z order, so the corner rows are served by the L2s: `traffic` (fabric bytes, PMC) is below the "
                    "algorithmic bytes; output rows are stored at a 272-byte stride (268 algorithmic bytes per row)"}


def gather_bwd_roofline(emb, log2_n=22, iters=10, warmup=3, frac_mode=0):
    """Table-gradient scatter (hm_encode_bwd_table): 12 + L*F*4 + L*8*F*4*2 algorithmic bytes/point
    (the RMW of all 8 corner rows is counted although zero-weight corners are skipped)."""
    from hashmodnffbanks_idr_amd import ops
    dev = emb.table.device
    n = 1 << log2_n
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = (torch.rand((n, 3), generator=g) * 2 - 1).to(dev)
    L, F = emb.n_levels, emb.n_features
    d_feat = torch.randn((n, L * F), device=dev)
    d_table = torch.zeros_like(emb.table)
    for _ in range(warmup):
        ops.encode_bwd_table(emb.desc, x, d_feat, frac_mode, out=d_table)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        ops.encode_bwd_table(emb.desc, x, d_feat, frac_mode, out=d_table)
        e.record()
    torch.cuda.synchronize()
    ms = np.asarray([s.elapsed_time(e) for s, e in evs])
    bpp = 12 + L * F * 4 + L * 8 * F * 4 * 2
    avg_ms = float(ms.mean())
    achieved = n * bpp / (avg_ms * 1e-3) / 1e9
    rec = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
           "kernel": "encode_bwd_table_zorder_kernel (+ 3 bucketing passes + the z-order pack pass, inside the timed launch)",
           "units_per_launch": n, "bytes_per_unit": bpp, "avg_launch_ms": round(avg_ms, 4),
           "frac_mode": "trilinear (all 8 corners carry weight)" if frac_mode else
                        "reference (only corner 0 carries weight: 1/8 of the read-modify-writes the byte count assumes)"}
    if frac_mode == 0:
        rec["all_corners"] = gather_bwd_roofline(emb, log2_n, max(2, iters // 2), 1, frac_mode=1)
    return rec


def mlp_roofline(net, log2_n=18, iters=10, warmup=4):
    """Fused SDF forward (encode + 9 MFMA layers), sdf-only output: 2*1 966 592 flop per point."""
    dev = next(net.parameters()).device
    n = 1 << log2_n
    g = torch.Generator(device="cpu").manual_seed(99)
    x = (torch.rand((n, 3), generator=g) * 2 - 1).to(dev)
    for _ in range(warmup):
        net.sdf(x)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        net.sdf(x)
        e.record()
    torch.cuda.synchronize()
    ms = np.asarray([s.elapsed_time(e) for s, e in evs])
    avg_ms = float(ms.mean())
    flops = 2.0 * (SDF_MAC_PER_POINT - 512 * 256)  # sdf-only: the 256 feature rows of the last layer are skipped
    tf = n * flops / (avg_ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": None, "kernel": "sdf_fwd_kernel",
            "units_per_launch": n, "flop_per_unit": flops, "avg_launch_ms": round(avg_ms, 4),
            "min_launch_ms": round(float(ms.min()), 4), "points_per_s": round(n / (avg_ms * 1e-3), 1)}


def mfma_stream_ceiling(device, iters=4000, reps=5):
    """What a stream of v_mfma_f32_32x32x2_f32 sustains on this GPU (hm_diag_mfma_f32_stream: the 64-point kernel's
    MFMA pattern - 8 waves per CU, four accumulators per wave - with operands in registers and no memory traffic)."""
    from hashmodnffbanks_idr_amd import _lib
    wgs = 256
    out = torch.empty(wgs * 512, dtype=torch.float32, device=device)
    call = lambda: _lib.check(_lib.lib().hm_diag_mfma_f32_stream(wgs, iters, _lib.dptr(out), _lib.stream_ptr(out)))  # noqa: E731
    call()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        call()
        e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e))
    flop = wgs * 8 * iters * 16 * 4096.0
    tf = flop / (min(ms) * 1e-3) / 1e12
    return {"TFLOP/s": round(tf, 2), "frac_of_nominal_peak": round(tf / MFMA_F32_PEAK_TF, 4),
            "what": "register-resident v_mfma_f32_32x32x2_f32 stream, 8 waves per CU, 4 accumulators per wave, "
                    f"{iters * 16} MFMAs per wave: the matrix pipe's own ceiling under this instruction (clock under load, "
                    "pipe occupancy per MFMA)"}


def gemm_roofline(device, iters=20, warmup=3):
    """Exact-fp32 GEMM of the grad path (csrc/hm_gemm.hip) on the shapes one training step runs: forward X W^T (+ Softplus
    epilogue) and input gradient dY W at 3072 / 2048 rows (~64 launches per step), and the weight gradients
    [u; z-bar]^T [v-bar; a] of the 8 hidden layers of one backward pass as ONE grouped launch (hm_gemm_f32_group_tn; 3 such
    launches per step).  `achieved` = flops of the whole list / its time."""
    from hashmodnffbanks_idr_amd import ops
    entries = []   # (label dict, callable, flops)
    for rows in (3072, 2048):
        for (M, N, K, ta, tb, sp) in ((rows, 512, 512, False, True, True), (rows, 512, 512, False, False, False)):
            a = torch.randn((K, M) if ta else (M, K), device=device)
            b = torch.randn((N, K) if tb else (K, N), device=device)
            out = torch.empty(M, N, device=device)
            bias = torch.zeros(N, device=device) if sp else None
            if sp:
                fn = (lambda a=a, b=b, bias=bias: ops.gemm_ep(a, b, bias, False, True, ops.EPI_SOFTPLUS, 100.0, 20.0))
            else:
                fn = (lambda a=a, b=b, ta=ta, tb=tb, out=out: ops.gemm(a, b, None, ta, tb, out=out))
            entries.append(({"M": M, "N": N, "K": K, "transA": ta, "transB": tb, "softplus_epilogue": sp}, fn,
                            2.0 * M * N * K))
        probs = [(torch.randn(2 * rows, 512, device=device), torch.randn(2 * rows, 512, device=device),
                  torch.zeros(512, 512, device=device)) for _ in range(8)]
        entries.append(({"grouped": 8, "M": 512, "N": 512, "K": 2 * rows, "transA": True, "transB": False,
                         "softplus_epilogue": False}, (lambda probs=probs: ops.gemm_group_tn(probs)),
                        8 * 2.0 * 512 * 512 * 2 * rows))
    per_shape, flops = [], 0.0
    side = torch.cuda.Stream()
    for label, fn, fl in entries:
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        # `iters` launches replayed from one HIP graph: the host cost of a launch (output allocation, ctypes call)
        # is out of the picture, as it is in the captured training step
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        g.replay()
        e0.record()
        torch.cuda.synchronize()
        us = s0.elapsed_time(e0) / iters * 1e3
        per_shape.append(dict(label, us=round(us, 2), **{"TFLOP/s": round(fl / us / 1e6, 1)}))
        flops += fl
        del g
    total_us = sum(p["us"] for p in per_shape)
    tf = flops / total_us / 1e6
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": None,
            "kernel": "gemm_f32_pipe2_kernel / gemm_f32_pipe2_group_kernel (20 launches per entry replayed from a HIP "
                      "graph, launch gaps included)",
            "shapes": per_shape}


def mlp_bf16_roofline(net, log2_n=18, iters=5, warmup=2):
    """bf16 coarse-search variant of the fused SDF kernel: same flop count, dense bf16 MFMA peak (2.5 PFLOP/s)."""
    from hashmodnffbanks_idr_amd import ops
    dev = next(net.parameters()).device
    n = 1 << log2_n
    g = torch.Generator(device="cpu").manual_seed(99)
    x = (torch.rand((n, 3), generator=g) * 2 - 1).to(dev)
    net.bf16_coarse_search = True
    emb = net._hash_embedder()
    pk = net.packed_weights()
    run = lambda: ops.sdf_fwd_bf16(emb.desc, pk, x, emb.table.detach(), emb.freq_encoding.B)   # noqa: E731
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        run()
        e.record()
    torch.cuda.synchronize()
    ms = np.asarray([s.elapsed_time(e) for s, e in evs])
    avg_ms = float(ms.mean())
    flops = 2.0 * (SDF_MAC_PER_POINT - 512 * 256)
    tf = n * flops / (avg_ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
            "traffic": None, "kernel": "sdf_fwd_bf16_kernel", "units_per_launch": n, "flop_per_unit": flops,
            "avg_launch_ms": round(avg_ms, 4), "points_per_s": round(n / (avg_ms * 1e-3), 1),
            "note": "bound by the per-CU weight stream (1 KB of bf16 weights per 6 MFMAs), not by the matrix pipe"}


def mlp_split_roofline(net, kind, log2_n=18, iters=5, warmup=2):
    """split-operand variant of the fused SDF kernel (csrc/hm_sdf_split.hip): same flop count as the fp32 kernel; the
    matrix work is 3 MFMAs of the 16-bit pipe per product, priced against the dense 16-bit MFMA peak (2.5 PFLOP/s)"""
    from hashmodnffbanks_idr_amd import ops
    dev = next(net.parameters()).device
    n = 1 << log2_n
    g = torch.Generator(device="cpu").manual_seed(99)
    x = (torch.rand((n, 3), generator=g) * 2 - 1).to(dev)
    net.coarse_split = kind
    emb = net._hash_embedder()
    pk = net.packed_weights()
    run = lambda: ops.sdf_fwd_split(emb.desc, pk, x, emb.table.detach(), emb.freq_encoding.B)   # noqa: E731
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        run()
        e.record()
    torch.cuda.synchronize()
    ms = np.asarray([s.elapsed_time(e) for s, e in evs])
    avg_ms = float(ms.mean())
    flops = 2.0 * (SDF_MAC_PER_POINT - 512 * 256)
    tf = n * flops / (avg_ms * 1e-3) / 1e12
    net.coarse_split = None
    return {"bound": "mfma", "achieved": round(3 * tf, 2), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(3 * tf / 2500.0, 4), "traffic": None, "kernel": f"sdf_fwd_split_kernel<{kind}>",
            "units_per_launch": n, "flop_per_unit": flops, "executed_flop_per_unit": 3 * flops,
            "fp32_equivalent_TFLOP/s": round(tf, 2), "avg_launch_ms": round(avg_ms, 4),
            "points_per_s": round(n / (avg_ms * 1e-3), 1),
            "note": "achieved = EXECUTED 16-bit MFMA flops (three products per fp32-equivalent product); "
                    "fp32_equivalent_TFLOP/s = the network's flops / time, comparable with roofline_mlp"}


def cpu_baseline(model, n_rays=256, reps=2):
    """oracle/torch_ref.py (port of the reference's PyTorch path) on the host cores: the SAME
    parameters the GPU run ended with (so both see the same surface / amount of ray-marching work),
    same workload shape at a bounded ray count: forward + IDRLoss + backward."""
    from oracle import torch_ref as R
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    ref = R.RefIDR(model)
    ref.train()
    inp, gt = synthetic_batch(1234, n_rays, "cpu")
    loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    cores = torch.get_num_threads()

    def one():
        out = ref(inp)
        lo = loss_fn(out, gt)
        ref.zero_grad()
        lo["loss"].backward()

    # thread sweep on the same sample: with ~2000 small ops per step the all-core setting is not the fastest one
    sweep = []
    for nt in sorted({cores, min(cores, 32), min(cores, 16), min(cores, 8)}, reverse=True):
        torch.set_num_threads(nt)
        one()
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        sweep.append((nt, (time.perf_counter() - t0) / reps))
    best_threads, dt_sample = min(sweep, key=lambda t: t[1])
    # the reported value: the WHOLE batch of the headline leg at the best thread count (one warm-up + `reps` steps)
    inp, gt = synthetic_batch(1234, RAYS_PER_GPU, "cpu")
    torch.set_num_threads(best_threads)
    one()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    torch.set_num_threads(cores)
    # the reference runner pins torch to ONE thread (training/idr_train.py:21): time that configuration too, on a
    # smaller sample (SURVEY.md 8d asks for both)
    n1 = 32
    inp, gt = synthetic_batch(1234, n1, "cpu")
    torch.set_num_threads(1)
    try:
        one()
        t0 = time.perf_counter()
        one()
        dt1 = time.perf_counter() - t0
    finally:
        torch.set_num_threads(cores)
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(RAYS_PER_GPU / dt, 2), "unit": "rays/s", "cores": best_threads, "kind": "port",
            "config": f"the {RAYS_PER_GPU}-ray batch of the GPU run's headline leg (same rays seed, same parameters); "
                      f"thread count chosen on a {n_rays}-ray sample; single_thread: 32-ray sample",
            "sample": f"all {RAYS_PER_GPU} rays, {reps} fwd+loss+bwd steps of oracle/torch_ref.py (torch-CPU, "
                      f"{best_threads} threads = best of the sweep), {dt:.2f} s/step",
            "threads_sweep": [{"threads": nt, "rays_per_s": round(n_rays / t, 2)} for nt, t in sweep],
            "host_cores": cores, "cpu_model": cpu_model,
            "single_thread": {"value": round(n1 / dt1, 2), "unit": "rays/s", "cores": 1,
                              "sample": f"{n1} rays, 1 step, torch.set_num_threads(1) as the reference runner does, "
                                        f"{dt1:.2f} s/step"}}


def _build(cfg, device, lr):
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    torch.manual_seed(0)  # identical replicas on every rank: geometric init, seed 0
    model = IDRNetwork(idr_conf(cfg)).to(device)
    model.train()
    if os.environ.get("HM_FUSED_MLP_GRAD", "1") == "0":   # debugging switch: generic autograd route for gradient()
        model.implicit_network.use_fused_mlp_grad = False
    return model


def _make_reducer(model, world, no_graph):
    """gradient exchange of a data-parallel leg: static steps use parallel.StaticGradExchange (hash-table gradients as
    (row, value) lists, ~1 MB per rank instead of 40 / 224 MB, device work captured into the step's graphs, two
    collectives per step); HM_DP_SPARSE=0 or --no-graph: dense all-reduce of everything (parallel.GradAllReducer)"""
    from hashmodnffbanks_idr_amd import parallel
    if not (world > 1 or os.environ.get("HM_DIST_FORCE") == "1"):    # (HM_DIST_FORCE: single-rank RCCL rehearsal)
        return None
    if no_graph or os.environ.get("HM_DP_SPARSE", "1") == "0":
        return parallel.GradAllReducer(model.parameters())
    tables = []
    for net in (model.implicit_network, model.rendering_network):
        emb = getattr(getattr(net, "embed_model", None), "embedder_obj", None)
        emb = getattr(emb, "grid_enc", emb)          # filter-bank embedders own a hash grid too
        # (row lists pay for tables a step touches a small part of; a table of a few rows - the view-direction grid - is
        #  cheaper as a dense gradient in the flat bucket: its scatter runs in an LDS copy of the table, 46 -> 4 us)
        if (emb is not None and hasattr(emb, "grad_collector") and emb.frac_mode == "reference"
                and int(emb.desc.total_rows) > 65536):
            tables.append(emb)
    return parallel.StaticGradExchange(model.parameters(), tables=tables)


def _spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (nothing has
    touched the GPU yet: counting devices does not initialise it), relay its output, exit with its code."""
    import socket
    import subprocess
    env = os.environ.copy()
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and "HM_DIST_BACKEND" not in env:
        # fewer devices than ranks (rehearsal on a one-GPU box): RCCL refuses two ranks per device, gloo does not
        env["HM_DIST_BACKEND"] = "gloo"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def _side_leg(cfg, device, bf16, steps=6, warmup=3, sampler_head=0, split=None):
    """short fixed-weights leg of another BASELINE configuration on rank 0 (reported beside the headline)"""
    import types
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss
    from hashmodnffbanks_idr_amd.training.optim import ClipAdam
    rays = NFFB_CONFIGS[cfg][5]
    a = types.SimpleNamespace(no_graph=False, warmup=warmup, steps=steps, rays=rays)
    model = _build(cfg, device, 0.0)
    model.implicit_network.bf16_coarse_search = bool(bf16)
    model.implicit_network.coarse_split = split
    model.ray_tracer.sampler_head = sampler_head      # 0, like the headline: the reference's evaluation count
    inp, gt = synthetic_batch(1234, rays, device)
    torch.manual_seed(100)
    opt, lfn = ClipAdam(model.parameters(), lr=0.0, max_norm=1.0), IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    dt, stats, final_loss, mode = _run_leg(a, model, opt, lfn, None, inp, gt, 1, device, 0, blocks=3)
    et, L, T = NFFB_CONFIGS[cfg][0], NFFB_CONFIGS[cfg][1], NFFB_CONFIGS[cfg][2]
    if sampler_head:
        return {"value": round(rays * steps / dt, 1), "unit": "rays/s", "ms_per_step": round(dt / steps * 1e3, 3),
                "sampler_head": sampler_head, "sdf_evals_per_step": stats}
    return {"value": round(rays * steps / dt, 1), "unit": "rays/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps, "timing": f"median of 3 consecutive blocks of {steps} steps", "rays": rays,
            "workload": f"{et} embedder L={L} T=2^{T} F=2, {rays} rays, weights at init (lr = 0)",
            "dtype": (f"{split} split operands (hi + lo 16-bit floats on the 16-bit MFMA, fp32 accumulate) in the coarse "
                      "ray-search scans, f32 elsewhere" if split else
                      "bf16 coarse ray-search scans, f32 elsewhere" if bf16 else "f32"),
            "sdf_evals_per_step": stats, "step": mode}


def _run_leg(args, model, opt, loss_fn, reducer, inp, gt, world, device, rank, blocks=1):
    """W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize; max over ranks.
    blocks > 1 (the short side legs only, never the headline): that many consecutive blocks of K timed steps, the MEDIAN
    block is reported - one transient stall in a 6-step block moved a side leg from 9.8 to 14.2 ms once (r3al)."""
    from hashmodnffbanks_idr_amd import parallel
    if args.no_graph:
        stepper = None

        def run_step():
            return parallel.train_step(model, loss_fn, opt, inp, gt, reducer)
    else:
        from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
        stepper = GraphedTrainStep(model, loss_fn, opt, reducer, warmup=2)

        def run_step():
            return stepper.step(inp, gt)
        for _ in range(max(0, 3 - args.warmup)):  # the capture must not fall into the timed region
            run_step()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    # tracer statistics of every timed step, kept on the device (no host read inside the timed region)
    hist = torch.zeros((args.steps, 16), dtype=torch.int32, device=device)
    rt = model.ray_tracer
    dts = []
    for _b in range(max(1, blocks)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            _, lo = run_step()
            if rt._stats_dev is not None:
                hist[i].copy_(rt._stats_dev, non_blocking=True)
        barrier()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2]
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    h = hist.cpu().numpy().astype(np.int64)
    ev = h[:, 6]
    stats = {"rays": args.rays, "mean": round(float(ev.mean()), 1), "min": int(ev.min()), "max": int(ev.max()),
             "per_ray_mean": round(float(ev.mean()) / args.rays, 2), "sampler_rays_mean": round(float(h[:, 0].mean()), 1),
             "mask_loss_rays_mean": round(float(h[:, 3].mean()), 1), "unfinished_max": int(h[:, 7].max()),
             "nonfinite_sdf_max": int(h[:, 8].max())}
    mode = ("eager (reference-structured, dynamic shapes)" if args.no_graph
            else ("HIP-graph captured static-shape step" if stepper.g_fb is not None
                  else "eager static-shape step (graph capture unavailable)"))
    return dt, stats, float(lo["loss"].item()), mode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU, help="rays per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline sections")
    ap.add_argument("--no-graph", action="store_true",
                    help="eager reference-structured step (dynamic shapes) instead of the HIP-graph captured step")
    ap.add_argument("--torch-adam", action="store_true",
                    help="torch.nn.utils.clip_grad_norm_ + torch.optim.Adam instead of the fused training.optim.ClipAdam")
    ap.add_argument("--legs", choices=["both", "fixed", "train", "lazy"], default="both",
                    help="fixed = section-8(d) workload (weights stay at geometric init, every sampler ray evaluates all "
                         "n_steps samples like the reference; the headline); train = the same with lr 1e-4; lazy = fixed "
                         "weights with the product's default lazy sampler (RayTracing.sampler_head = 16: samples past a "
                         "ray's first sign change are not evaluated, outputs bit-identical); both = all three")
    ap.add_argument("--gather-log2n", type=int, default=22)
    ap.add_argument("--calib", default="1,0", help="gather_calib: lanes per 128-B block, byte stride between them")
    ap.add_argument("--bf16", type=int, default=-1,
                    help="1: the ray tracer's coarse scans on the bf16 kernel (default for --cfg C5 = BASELINE configs[4])")
    ap.add_argument("--split", choices=["bf16x2", "f16x2"], default=None,
                    help="coarse scans of every leg on the split-operand kernel (default for --cfg C5: bf16x2)")
    ap.add_argument("--only", choices=["gather", "gather_bwd", "mlp", "mlp_bf16", "mlp_split", "gemm", "gather_calib"], default=None,
                    help="profiling helper: run just one kernel section on cuda:0 and print its object")
    ap.add_argument("--cfg", default=None,
                    help="hash-grid config (tests/golden/params.py): default C2 = BASELINE configs[1] at every --gpus N; "
                         "C4 = configs[3] (T = 2^22), C3 / C5 = the filter-bank configurations")
    args = ap.parse_args()
    cfg = args.cfg or os.environ.get("HM_BENCH_CFG") or CFG
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and not args.only:
        sys.exit(_spawn_ranks(args))

    if args.only == "gather_calib":
        # PMC calibration of 8-byte gathers (run under rocprofv3 --pmc ...; see profiles/README.md)
        from hashmodnffbanks_idr_amd import _lib
        dev = torch.device("cuda", 0)
        group, stride = (int(v) for v in args.calib.split(","))
        table = torch.zeros((4 << 30) // 4, dtype=torch.float32, device=dev)      # 4 GiB >> 256 MiB Infinity Cache
        n = 1 << 24
        out = torch.empty(n, dtype=torch.float32, device=dev)
        for _ in range(5):
            _lib.check(_lib.lib().hm_diag_gather_calib(_lib.dptr(table), table.numel() * 4, n, group, stride,
                                                       _lib.dptr(out), _lib.stream_ptr(out)))
        torch.cuda.synchronize()
        print(json.dumps({"kernel": "gather_calib_kernel", "lanes": n, "group": group, "stride_bytes": stride,
                          "blocks_per_launch": n // group, "rows_bytes_per_launch": n * 8, "launches": 5}))
        return
    if args.only:
        dev = torch.device("cuda", 0)
        model = _build(cfg, dev, 0.0)
        emb = model.implicit_network.embed_model.embedder_obj
        if args.only == "gather":
            print(json.dumps(gather_roofline(emb, args.gather_log2n)))
        elif args.only == "gather_bwd":
            print(json.dumps(gather_bwd_roofline(emb, args.gather_log2n)))
        elif args.only == "mlp_bf16":
            print(json.dumps(mlp_bf16_roofline(model.implicit_network)))
        elif args.only == "mlp_split":
            print(json.dumps(mlp_split_roofline(model.implicit_network, args.split or "f16x2")))
        elif args.only == "gemm":
            print(json.dumps(gemm_roofline(dev)))
        else:
            r = mlp_roofline(model.implicit_network)
            r["mfma_stream_ceiling"] = mfma_stream_ceiling(dev)
            r["frac_of_stream_ceiling"] = round(r["achieved"] / r["mfma_stream_ceiling"]["TFLOP/s"], 4)
            print(json.dumps(r))
        return

    from hashmodnffbanks_idr_amd import parallel
    from hashmodnffbanks_idr_amd.model.loss import IDRLoss

    rank, world, local_rank = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    device = torch.device("cuda", local_rank % max(n_dev, 1))  # (rehearsal: several gloo ranks may share one GPU)
    torch.cuda.set_device(device)
    loss_fn = IDRLoss(eikonal_weight=0.1, mask_weight=100.0, alpha=50.0)
    inp, gt = synthetic_batch(1234 + rank, args.rays, device)
    if cfg in NFFB_CONFIGS:
        L, T = NFFB_CONFIGS[cfg][1], NFFB_CONFIGS[cfg][2]
        if args.rays == RAYS_PER_GPU:
            args.rays = NFFB_CONFIGS[cfg][5]
            inp, gt = synthetic_batch(1234 + rank, args.rays, device)
    else:
        L, T = P.CONFIGS[cfg][0], P.CONFIGS[cfg][1]

    def make_opt(model, lr):
        # dense Adam over ALL parameters incl. the hash table (reference: idr_train.py:127-128)
        if args.torch_adam:
            return torch.optim.Adam(model.parameters(), lr=lr, capturable=not args.no_graph)
        from hashmodnffbanks_idr_amd.training.optim import ClipAdam   # clip_grad_norm_(1.0) + Adam in three launches
        return ClipAdam(model.parameters(), lr=lr, max_norm=1.0)

    use_split = args.split or ("bf16x2" if (cfg == "C5" and args.bf16 < 0) else None)
    use_bf16 = (args.bf16 == 1) and not use_split

    def run_one(cfg_, lr, sampler_head, steps, warmup, bf16, split=None):
        """one timed leg on ALL ranks (it contains collectives when world > 1)"""
        import types
        a = types.SimpleNamespace(no_graph=args.no_graph, warmup=warmup, steps=steps, rays=args.rays)
        model = _build(cfg_, device, lr)
        model.implicit_network.bf16_coarse_search = bf16
        model.implicit_network.coarse_split = split
        model.ray_tracer.sampler_head = sampler_head
        reducer = _make_reducer(model, world, args.no_graph)
        torch.manual_seed(100 + rank)  # per-rank eikonal points / step fractions
        dt, stats, final_loss, mode = _run_leg(a, model, make_opt(model, lr), loss_fn, reducer, inp, gt, world, device, rank)
        rec = {"value": round(args.rays * world * steps / dt, 1), "unit": "rays/s",
               "ms_per_step": round(dt / steps * 1e3, 3), "lr": lr, "sampler_head": sampler_head,
               "sdf_evals_per_step": stats, "final_loss": round(final_loss, 6), "step": mode}
        if reducer is not None:
            rec["exchange"] = type(reducer).__name__
            if hasattr(reducer, "check"):
                reducer.check()        # (host read after the timed region: no touched row may have missed its payload)
        return rec, model

    legs = {}
    model = None
    for leg, lr, sampler_head in (("fixed", 0.0, 0), ("train", 1.0e-4, 0), ("lazy", 0.0, LAZY_SAMPLER_HEAD)):
        if args.legs not in ("both", leg):
            continue
        legs[leg], model = run_one(cfg, lr, sampler_head, args.steps, args.warmup, use_bf16, split=use_split)
        if leg == "fixed":
            head_model = model
    if "fixed" not in legs:
        head_model = model
    split_leg = None
    if not args.no_extras and cfg in ("C2", "C4") and args.legs in ("both", "fixed") and world == 1:
        # the headline iteration with the coarse scans on the split-operand kernel (fp16 hi + lo, 22-bit operands):
        # same workload, same evaluation count; reported beside the headline, which stays exact fp32
        split_leg, _m = run_one(cfg, 0.0, 0, args.steps, args.warmup, False, split="f16x2")
        split_leg["coarse_scans"] = "f16x2 split operands on v_mfma_f32_32x32x16_f16 (csrc/hm_sdf_split.hip)"
        del _m
    c4_leg = None
    if not args.no_extras and cfg == "C2" and args.rays == RAYS_PER_GPU:
        # BASELINE configs[3] (T = 2^22; 16 384 rays over 8 GPUs = these 2048 rays per GPU), short fixed-weights leg at
        # EVERY N so that its own 1 -> 8 curve exists beside the headline's
        c4_leg, _m = run_one("C4", 0.0, 0, min(args.steps, 10), 3, False)
        c4_leg["workload"] = "MultiResHash L=16 T=2^22 F=2 (BASELINE.json configs[3]), 2048 rays per GPU, weights at init"
        c4_leg["steps"] = min(args.steps, 10)
        del _m

    if rank == 0:
        head = legs.get("fixed") or legs.get("train") or legs["lazy"]
        line = {
            "metric": "rays/sec fwd+bwd (hash+SDF MLP)", "value": head["value"], "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "data": "synthetic",
            "dtype": (f"{use_split} (ray-search coarse scans with hi + lo 16-bit split operands on the 16-bit MFMA, fp32 "
                      "accumulate; sphere tracing, secant refinement and every grad-enabled evaluation f32)" if use_split else
                      "bf16 (ray-search coarse scans on v_mfma_f32_32x32x16_bf16, fp32 accumulate; sphere tracing, secant "
                      "refinement and every grad-enabled evaluation f32)" if use_bf16 else "f32"),
            "config": {"workload": f"DTU-shaped synthetic uniform-sphere rays, "
                                   + (f"{NFFB_CONFIGS[cfg][0]} embedder over a hash grid " if cfg in NFFB_CONFIGS else "MultiResHash ")
                                   + f"L={L} T=2^{T} F=2 "
                                   f"(BASELINE.json configs[{ {'C4': 3, 'C3': 2, 'C5': 4}.get(cfg, 1) }]), full IDR training step "
                                   "(forward + IDRLoss + backward + clip + Adam)"
                                   + (", weights held at the geometric initialisation (lr = 0): the SURVEY.md 8(d) workload"
                                      if "fixed" in legs else ", lr 1e-4 (surface moves during the timed region)"),
                       "grid_config": cfg, "rays_per_gpu": args.rays, "global_rays": args.rays * world,
                       "parallelism": f"ray-sharded dp{world}" if world > 1 else "single GPU",
                       "world": world,
                       "backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
                       "ranks_seen": (torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1),
                       "exchange": head.get("exchange"),
                       "step": head["step"],
                       "sampler": ("all n_steps samples of every unconverged ray are evaluated, as in the reference "
                                   "(ray_tracing.py:189-249)" if head["sampler_head"] == 0 else
                                   f"lazy: first pass over samples 0..{head['sampler_head'] - 1} + the last, second pass "
                                   "for rays without a sign change there; outputs bit-identical to the single pass"),
                       "sdf_evals_per_step": head["sdf_evals_per_step"]},
            "final_loss": head["final_loss"],
        }
        if "fixed" in legs and "train" in legs:
            line["train_leg"] = legs["train"]
        if "lazy" in legs and head is not legs["lazy"]:
            # same iteration, same weights, same outputs (tests/test_raytrace_gpu.py::test_lazy_sampler_equals_single_pass);
            # the product default.  Reported beside the headline because SURVEY.md 8(d) characterises the workload by
            # the reference's evaluation count (~123 per ray), which the lazy sampler undercuts.
            line["lazy_sampler_leg"] = legs["lazy"]
        if c4_leg is not None:
            line["config4_leg"] = c4_leg
        if split_leg is not None:
            line["split_f16x2_leg"] = split_leg
        if not args.no_extras and world > 1:
            # N > 1: the kernel rooflines / CPU baseline are single-GPU measurements and belong to the N = 1 line
            emb = head_model.implicit_network.embed_model.embedder_obj
            emb = getattr(emb, "grid_enc", emb)
            if not (cfg in NFFB_CONFIGS):
                line["roofline"] = gather_roofline(emb, args.gather_log2n)
        elif not args.no_extras and cfg in NFFB_CONFIGS:
            line["cpu_baseline"] = None      # (the torch-CPU port covers the hash-grid embedder only)
        elif not args.no_extras:
            emb = head_model.implicit_network.embed_model.embedder_obj
            line["roofline"] = gather_roofline(emb, args.gather_log2n)
            line["roofline_bwd"] = gather_bwd_roofline(emb, args.gather_log2n, iters=5, warmup=2)
            line["roofline_mlp"] = mlp_roofline(head_model.implicit_network)
            # the same MFMA stream without memory traffic: how much of the gap to the nominal peak is the pipe's own
            ceil = mfma_stream_ceiling(device)
            line["roofline_mlp"]["mfma_stream_ceiling"] = ceil
            line["roofline_mlp"]["frac_of_stream_ceiling"] = round(line["roofline_mlp"]["achieved"] / ceil["TFLOP/s"], 4)
            if cfg != "C4":
                from hashmodnffbanks_idr_amd.model.embeddings.hashGridEmbedding import MultiResHashGridMLP
                Lc, Tc, bc, dc = P.CONFIGS["C4"]
                torch.manual_seed(0)
                emb4 = MultiResHashGridMLP(True, 3, Lc, 2, Tc, bc, dc).to(device)
                line["roofline_c4"] = gather_roofline(emb4, args.gather_log2n)
                del emb4
            line["roofline_mlp_bf16"] = mlp_bf16_roofline(_build(cfg, device, 0.0).implicit_network)
            line["roofline_mlp_f16x2"] = mlp_split_roofline(_build(cfg, device, 0.0).implicit_network, "f16x2")
            line["roofline_mlp_bf16x2"] = mlp_split_roofline(_build(cfg, device, 0.0).implicit_network, "bf16x2")
            line["roofline_gemm"] = gemm_roofline(device)
            line["cpu_baseline"] = cpu_baseline(head_model)
            if world == 1:     # BASELINE configs[2] and [4] (filter-bank embedders), short legs beside the headline
                # (sampler_head = 16 changes nothing at these networks' initialisation: none of their sampler rays
                #  has a sign change among the head samples, both passes run in full - 33.0 vs 32.5 ms at C3)
                line["config3_leg"] = _side_leg("C3", device, False)
                # configs[4] ("bf16"): the coarse scans with bf16 hi + lo split operands (kernel error 5e-6 against the fp32
                # kernel; its training curve stays with the fp32 run's through steps 0 - 9 in every run and through 10 - 19 in
                # most - this network's training is chaotic -, tests/test_split_gpu.py) and,
                # beside it, with plain bf16 operands (1.2e-3; leaves the fp32 curve after ~10 steps, tests/test_bf16_gpu.py)
                line["config5_leg"] = _side_leg("C5", device, False, split="bf16x2")
                line["config5_leg_plain_bf16"] = _side_leg("C5", device, True)
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
