"""RayTracing - IDR's ray/surface intersection search (plugin point #3), same constructor and
``forward(sdf, cam_loc, object_mask, ray_directions) -> (points, network_object_mask, dists)``
contract as the reference (code/model/ray_tracing.py:6-95).

Restated as a per-ray state machine over flat [N] state vectors (every update in the reference
is gated by a per-ray mask, so the two formulations are equivalent - SURVEY.md Appendix C (v)):

  1. bidirectional sphere tracing with line-search back-off    (reference :98-187)
  2. 100-sample sign-change search + 8 secant steps for rays that did not converge (:189-268)
  3. training only: closest-approach search for mask-loss rays (:270-298)

Differences in mechanism, not behaviour: the near and far marches of one iteration are evaluated
in ONE batched SDF call (the fused HIP kernel is tile-parallel, so a second launch would only
add latency); the 10 000-point chunking of the reference is gone (the kernel grid-strides over
any N); the 100 random fractions of step 3 come from the same global CPU generator call the
reference makes (``torch.empty(n).uniform_(0, 1)``) unless ``steps`` is injected.
"""
import torch
import torch.nn as nn

from .. import _lib, ops
from ..utils import rend_util


class RayTracing(nn.Module):
    def __init__(self, object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=1,
                 sphere_tracing_iters=10, n_steps=100, n_secant_steps=8):
        super().__init__()
        self.object_bounding_sphere = object_bounding_sphere
        self.sdf_threshold = sdf_threshold
        self.sphere_tracing_iters = sphere_tracing_iters
        self.line_step_iters = line_step_iters
        self.line_search_step = line_search_step
        self.n_steps = n_steps
        self.n_secant_steps = n_secant_steps
        self.verbose = False      # the reference prints three lines per call (:61-64), each a device sync
        self.steps_override = None  # optional [n_steps] tensor replacing the U(0,1) draw of step 3
        self.sphere_override = None  # optional (t_sphere [B,N,2], hit [B,N]) the caller already has (ops.camera_rays)
        self.use_device_tracer = True   # sync-free HIP state-machine tracer when `sdf` is the package's network
        # device tracer: the sampler's first pass evaluates samples 0..sampler_head-1 (and the last one), the second
        # pass the remaining samples of the rays whose first sign change is not among them - the reference reads
        # nothing past a ray's first negative sample (:212-218), so the outputs are bit-identical to evaluating all
        # n_steps samples of every ray (sampler_head = 0, what the reference and the generic path below do)
        self.sampler_head = 16
        self._stats = {}
        self._stats_dev = None
        self._ws = None

    @property
    def last_stats(self):
        """Counters of the last call (reads them back from the device on access)."""
        if self._stats_dev is not None:
            v = self._stats_dev.tolist()
            # (the device tensor is kept: inside a captured graph it is refreshed by every replay)
            self._stats = {"rays": self._stats.get("rays"), "sampler_rays": v[0], "secant_rays": v[2],
                           "mask_loss_rays": v[3], "sdf_evals": v[6], "unfinished": v[7], "nonfinite": v[8],
                           "sampler_points": v[1], "sampler_second_pass_rays": v[10]}
        return self._stats

    def _fused_network(self, sdf, ray_directions):
        net = getattr(sdf, "__self__", None)
        if (self.use_device_tracer and net is not None and ray_directions.is_cuda and hasattr(net, "packed_weights")
                and getattr(sdf, "__func__", None) is getattr(type(net), "sdf", None) and net._fusable()):
            return net
        return None

    def _forward_device(self, net, cam_loc, object_mask, ray_directions):
        """The whole search enqueued by ONE C-ABI call (csrc/hm_trace.hip); no host synchronisation."""
        B, P, _ = ray_directions.shape
        N = B * P
        dev = ray_directions.device
        with torch.no_grad():
            if self.sphere_override is not None:
                t_sphere, hit = self.sphere_override
            else:
                t_sphere, hit = rend_util.get_sphere_intersection(cam_loc.detach(), ray_directions.detach(),
                                                                  r=self.object_bounding_sphere)
            cfg = _lib.TraceCfg(float(self.object_bounding_sphere), float(self.sdf_threshold),
                                float(self.line_search_step), int(self.line_step_iters),
                                int(self.sphere_tracing_iters), int(self.n_steps), int(self.n_secant_steps),
                                1 if self.training else 0,
                                net.coarse_mode() if hasattr(net, "coarse_mode") else 0,
                                int(self.sampler_head))
            nf = net._nffb_embedder() if net._hash_embedder() is None else None
            need = ops.trace_workspace_bytes(N, cfg, nf.n_levels if nf is not None else 0)
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            steps_u = None
            if self.training:
                if self.steps_override is not None:
                    steps_u = self.steps_override.to(dev).float().contiguous()
                else:
                    # same generator call as the reference (:277); it is made even when no ray ends up needing it
                    steps_u = torch.empty(self.n_steps).uniform_(0.0, 1.0).to(dev)
            emb = net._hash_embedder() if nf is None else nf.grid_enc
            nffb = ops.nffb_packed(nf) if nf is not None else None
            stats = torch.zeros(16, dtype=torch.int32, device=dev)
            pts, mask, dists = ops.trace_forward(
                emb.desc, net.packed_weights(), emb.table.detach(), emb.freq_encoding.B,
                ops.FRAC_MODES[emb.frac_mode], net.sdf_tile_points, cfg, cam_loc.detach().contiguous().float(),
                ray_directions.detach().reshape(N, 3).contiguous().float(),
                object_mask.reshape(N).to(torch.uint8).contiguous(), t_sphere.reshape(N, 2).contiguous(),
                hit.reshape(N).to(torch.uint8).contiguous(), P, self._linspace(dev), steps_u, self._ws, stats,
                nffb=nffb)
        self._stats = {"rays": N}
        self._stats_dev = stats
        return pts, mask.bool(), dists

    # ------------------------------------------------------------------------------------
    def forward(self, sdf, cam_loc, object_mask, ray_directions):
        net = self._fused_network(sdf, ray_directions)
        if net is not None:
            return self._forward_device(net, cam_loc, object_mask, ray_directions)
        B, P, _ = ray_directions.shape
        N = B * P
        dirs = ray_directions.reshape(N, 3)
        cams = cam_loc.unsqueeze(1).expand(B, P, 3).reshape(N, 3)

        t_sphere, hit = rend_util.get_sphere_intersection(cam_loc, ray_directions, r=self.object_bounding_sphere)
        t_sphere = t_sphere.reshape(N, 2)
        hit = hit.reshape(N)

        pts, unfinished, t_start, t_end, t_min, t_max = self._sphere_trace(sdf, cams, dirs, hit, t_sphere)
        net_mask = t_start < t_end

        sampler_mask = unfinished
        n_sampler = int(sampler_mask.sum())
        n_secant_hits = 0
        if n_sampler > 0:
            idx = torch.nonzero(sampler_mask).flatten()
            s_pts, s_hit, s_t = self._sample_and_secant(sdf, cams[idx], dirs[idx], t_start[idx], t_end[idx],
                                                        object_mask[idx])
            pts[idx] = s_pts
            t_start[idx] = s_t
            net_mask[idx] = s_hit
            n_secant_hits = int(s_hit.sum())
        self._stats, self._stats_dev = {"rays": N, "sampler_rays": n_sampler}, None
        if self.verbose:
            print('----------------------------------------------------------------')
            print('RayTracing: object = {0}/{1}, secant on {2}/{3}.'.format(int(net_mask.sum()), N, n_secant_hits,
                                                                           n_sampler))
            print('----------------------------------------------------------------')

        if not self.training:
            return pts, net_mask, t_start

        in_mask = ~net_mask & object_mask & ~sampler_mask
        out_mask = ~object_mask & ~sampler_mask
        loss_rays = in_mask | out_mask

        missed = loss_rays & ~hit
        if bool(missed.any()):
            # rays that never enter the sphere: closest point of the ray to the origin
            c, d = cams[missed], dirs[missed]
            t_proj = -torch.bmm(d.view(-1, 1, 3), c.view(-1, 3, 1)).squeeze()
            t_start[missed] = t_proj
            pts[missed] = c + t_start[missed].unsqueeze(1) * d

        sel = loss_rays & hit
        if bool(sel.any()):
            moved = net_mask & out_mask
            t_min[moved] = t_start[moved]
            m_pts, m_t = self._closest_approach(sdf, cams[sel], dirs[sel], t_min[sel], t_max[sel])
            pts[sel] = m_pts
            t_start[sel] = m_t
        return pts, net_mask, t_start

    # ------------------------------------------------------------------------------------
    def _eval_masked(self, sdf, pts_a, mask_a, pts_b, mask_b, out_a, out_b):
        """out_x[mask_x] = sdf(pts_x[mask_x]) for both sides with one SDF launch."""
        ia = torch.nonzero(mask_a).flatten()
        ib = torch.nonzero(mask_b).flatten()
        na = ia.numel()
        if na + ib.numel() == 0:
            return
        vals = sdf(torch.cat([pts_a[ia], pts_b[ib]], 0))
        out_a[ia] = vals[:na]
        out_b[ib] = vals[na:]

    def _sphere_trace(self, sdf, cams, dirs, hit, t_sphere):
        """March from both sphere intersections towards each other (reference :98-187)."""
        N = dirs.shape[0]
        dev = dirs.device
        thr = self.sdf_threshold
        zeros = torch.zeros(N, device=dev, dtype=torch.float32)

        def along(t):
            return cams + t.unsqueeze(-1) * dirs

        live_s, live_e = hit.clone(), hit.clone()
        t_s = torch.where(hit, t_sphere[:, 0], zeros)
        t_e = torch.where(hit, t_sphere[:, 1], zeros)
        p_s = torch.where(hit.unsqueeze(-1), along(t_sphere[:, 0]), torch.zeros_like(dirs))
        p_e = torch.where(hit.unsqueeze(-1), along(t_sphere[:, 1]), torch.zeros_like(dirs))
        t_min, t_max = t_s.clone(), t_e.clone()

        nxt_s, nxt_e = zeros.clone(), zeros.clone()
        self._eval_masked(sdf, p_s, live_s, p_e, live_e, nxt_s, nxt_e)

        it = 0
        while True:
            cur_s = torch.where(live_s, nxt_s, zeros)
            cur_s = torch.where(cur_s <= thr, zeros, cur_s)
            cur_e = torch.where(live_e, nxt_e, zeros)
            cur_e = torch.where(cur_e <= thr, zeros, cur_e)
            live_s = live_s & (cur_s > thr)
            live_e = live_e & (cur_e > thr)
            if it == self.sphere_tracing_iters or not bool((live_s | live_e).any()):
                break
            it += 1

            t_s = t_s + cur_s
            t_e = t_e - cur_e
            p_s, p_e = along(t_s), along(t_e)

            nxt_s, nxt_e = zeros.clone(), zeros.clone()
            self._eval_masked(sdf, p_s, live_s, p_e, live_e, nxt_s, nxt_e)

            # a step that landed inside the surface is pulled back (halving) up to line_step_iters times
            over_s, over_e = nxt_s < 0, nxt_e < 0
            k = 0
            while k < self.line_step_iters and bool((over_s | over_e).any()):
                back = (1 - self.line_search_step) / (2 ** k)
                t_s = torch.where(over_s, t_s - back * cur_s, t_s)
                t_e = torch.where(over_e, t_e + back * cur_e, t_e)
                p_s = torch.where(over_s.unsqueeze(-1), along(t_s), p_s)
                p_e = torch.where(over_e.unsqueeze(-1), along(t_e), p_e)
                self._eval_masked(sdf, p_s, over_s, p_e, over_e, nxt_s, nxt_e)
                over_s, over_e = nxt_s < 0, nxt_e < 0
                k += 1

            crossed = t_s < t_e
            live_s = live_s & crossed
            live_e = live_e & crossed
        return p_s, live_s, t_s, t_e, t_min, t_max

    def _linspace(self, dev):
        ls = getattr(self, "_ls_cache", None)
        if ls is None or ls.device != dev or ls.numel() != self.n_steps:
            # built on the CPU like the reference (:198) so the fp32 sample fractions are identical
            ls = torch.linspace(0, 1, steps=self.n_steps).to(dev)
            self._ls_cache = ls
        return ls

    def _sample_and_secant(self, sdf, cams, dirs, t0, t1, true_obj):
        """Uniform samples in [t0,t1], first sign change, secant refinement (reference :189-268).
        All arguments are already restricted to the M unconverged rays."""
        M = dirs.shape[0]
        dev = dirs.device
        n = self.n_steps
        frac = self._linspace(dev).view(1, n)
        ts = t0.unsqueeze(-1) + frac * (t1 - t0).unsqueeze(-1)                      # [M,n]
        pts = cams.unsqueeze(1) + ts.unsqueeze(-1) * dirs.unsqueeze(1)              # [M,n,3]
        vals = sdf(pts.reshape(-1, 3)).reshape(M, n)

        # first negative sample: argmin of sign(v) * (n, n-1, ..., 1)
        rank = torch.arange(n, 0, -1, device=dev, dtype=torch.float32).view(1, n)
        first = torch.argmin(torch.sign(vals) * rank, -1)
        rows = torch.arange(M, device=dev)
        out_pts = pts[rows, first]
        out_t = ts[rows, first]
        v_first = vals[rows, first]
        net_hit = v_first < 0

        # rays that are not (true surface & network surface): take the minimal-SDF sample instead
        p_out = ~(true_obj & net_hit)
        if bool(p_out.any()):
            io = torch.nonzero(p_out).flatten()
            amin = torch.argmin(vals[io], -1)
            out_pts[io] = pts[io, amin]
            out_t[io] = ts[io, amin]

        sec = (net_hit & true_obj) if self.training else net_hit
        if bool(sec.any()):
            isec = torch.nonzero(sec).flatten()
            f = first[isec]
            z_hi, v_hi = ts[isec, f], vals[isec, f]
            z_lo, v_lo = ts[isec, f - 1], vals[isec, f - 1]   # f == 0 wraps to the last sample, as in the reference
            z = self._secant(sdf, v_lo, v_hi, z_lo, z_hi, cams[isec], dirs[isec])
            out_pts[isec] = cams[isec] + z.unsqueeze(-1) * dirs[isec]
            out_t[isec] = z
        return out_pts, net_hit, out_t

    def _secant(self, sdf, v_lo, v_hi, z_lo, z_hi, cams, dirs):
        """n_secant_steps iterations of the secant rule on a bracketing pair (reference :251-268)."""
        z = -v_lo * (z_hi - z_lo) / (v_hi - v_lo) + z_lo
        for _ in range(self.n_secant_steps):
            v = sdf(cams + z.unsqueeze(-1) * dirs)
            pos, neg = v > 0, v < 0
            z_lo = torch.where(pos, z, z_lo)
            v_lo = torch.where(pos, v, v_lo)
            z_hi = torch.where(neg, z, z_hi)
            v_hi = torch.where(neg, v, v_hi)
            z = -v_lo * (z_hi - z_lo) / (v_hi - v_lo) + z_lo
        return z

    def _closest_approach(self, sdf, cams, dirs, t_lo, t_hi):
        """argmin of the SDF over n_steps shared random fractions of [t_lo, t_hi] (reference :270-298)."""
        M = dirs.shape[0]
        dev = dirs.device
        n = self.n_steps
        if self.steps_override is not None:
            u = self.steps_override.to(dev).float()
        else:
            u = torch.empty(n).uniform_(0.0, 1.0).to(dev)   # one draw shared by all rays, global CPU RNG (:277)
        ts = u.unsqueeze(0).repeat(M, 1) * (t_hi.unsqueeze(-1) - t_lo.unsqueeze(-1)) + t_lo.unsqueeze(-1)
        pts = cams.unsqueeze(1) + ts.unsqueeze(-1) * dirs.unsqueeze(1)
        vals = sdf(pts.reshape(-1, 3)).reshape(M, n)
        amin = torch.argmin(vals, -1)
        rows = torch.arange(M, device=dev)
        return pts[rows, amin], ts[rows, amin]
