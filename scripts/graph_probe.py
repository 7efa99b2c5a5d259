"""Which C-ABI calls survive HIP-graph capture?  Each probe runs in its own process."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES = ["stepper"]

def run(name):
    sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
    import torch, numpy as np
    import bench
    from helpers import idr_conf
    from hashmodnffbanks_idr_amd import ops
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    torch.manual_seed(0)
    model = IDRNetwork(idr_conf("C1")).cuda()
    net = model.implicit_network
    emb = net.embed_model.embedder_obj
    x = torch.rand(3000, 3, device="cuda") * 2 - 1
    a = torch.randn(1750, 512, device="cuda"); b = torch.randn(512, 512, device="cuda")
    a2 = torch.randn(3000, 512, device="cuda"); b2 = torch.randn(3000, 512, device="cuda")
    inp, gt = bench.synthetic_batch(1, 512, "cuda")
    eik = torch.rand(256, 3, device="cuda") * 2 - 1
    steps = torch.rand(100, device="cuda")
    def body():
        if name == "encode":
            return ops.encode_fwd(emb.desc, x, emb.table.detach(), emb.freq_encoding.B, 0)
        if name == "gemm":
            return ops.gemm(a, b, None, False, True)
        if name == "gemm_splitk":
            return ops.gemm(a2, b2, None, True, False)
        if name == "sdf16":
            net.sdf_tile_points = 16; return net.sdf(x)
        if name == "sdf64":
            net.sdf_tile_points = 64; return net.sdf(x)
        if name == "trace":
            from hashmodnffbanks_idr_amd.utils import rend_util
            rd, cl = rend_util.get_camera_params(inp["uv"], inp["pose"], inp["intrinsics"])
            model.ray_tracer.steps_override = torch.rand(100, device="cuda")
            with torch.no_grad():
                return model.ray_tracer(sdf=net.sdf, cam_loc=cl, object_mask=inp["object_mask"].reshape(-1), ray_directions=rd)
        if name == "encode_bwd":
            emb.table.grad = None
            y = emb(x)
            y.sum().backward()
            return emb.table.grad
        if name in ("fwg", "fwg_bwd"):
            model.zero_grad(set_to_none=True)
            xx = x.clone()
            out, g = net.forward_with_gradient(xx)
            if name == "fwg_bwd":
                ((g ** 2).sum() + out.sum()).backward()
            return out
        if name in ("fwd_static", "fwd_bwd"):
            from hashmodnffbanks_idr_amd.model.loss import idr_loss_terms as idr_loss_static
            model.zero_grad(set_to_none=True)
            model.train()
            out = model.forward_static(inp, eik, steps)
            lo = idr_loss_static(out, gt["rgb"], 0.1, 100.0, 50.0)
            if name == "fwd_bwd":
                lo["loss"].backward()
            return lo
        if name == "repack":
            net._packed = None
            return net.sdf(x)
        if name in ("fwd_bwd_repack", "fwd_bwd_defaultwarm"):
            from hashmodnffbanks_idr_amd.model.loss import idr_loss_terms as idr_loss_static
            model.zero_grad(set_to_none=True)
            model.train()
            net._packed = None
            out = model.forward_static(inp, eik, steps)
            lo = idr_loss_static(out, gt["rgb"], 0.1, 100.0, 50.0)
            lo["loss"].backward()
            return lo
        if name == "bwd_thread":
            w = torch.randn(512, 512, device="cuda", requires_grad=True)
            y = ops.linear(a, w, None)
            y.sum().backward()
            return w.grad
    if name.startswith("stepper"):
        from hashmodnffbanks_idr_amd.training.graph_step import GraphedTrainStep
        from hashmodnffbanks_idr_amd.model.loss import IDRLoss
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
        st = GraphedTrainStep(model, IDRLoss(0.1, 100.0, 50.0), opt, warmup=2)
        if name == "stepper_nograph_opt":
            st._update = lambda: None
        for i in range(5):
            out, lo = st.step(inp, gt)
            print("step", i, lo["loss"].item(), flush=True)
        print("PROBE", name, "ok", flush=True)
        return
    if name == "fwd_bwd_defaultwarm":
        for _ in range(2):
            r = body()
    else:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2):
                r = body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r = body()
    g.replay(); g.replay()
    torch.cuda.synchronize()
    print("PROBE", name, "ok", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for p in PROBES:
            rc = subprocess.call([sys.executable, os.path.abspath(__file__), p], stdout=sys.stdout, stderr=subprocess.DEVNULL)
            print("PROBE", p, "exit code", rc, flush=True)
