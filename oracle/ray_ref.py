"""torch-CPU restatement of the reference's ray / surface intersection search and of the small elementwise
pieces around it (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

Independent of the product: nothing here imports hashmodnffbanks_idr_amd.  Where the product runs per-ray state
machines, this file follows the reference's *batch* formulation (full-length [N] state vectors, boolean masks),
written from the algorithm description with its own helper structure; citations are to
/root/reference/code/model/ray_tracing.py, utils/rend_util.py and model/sample_network.py.

Parity status: PINNED by tests/test_oracle_golden.py::test_ray_ref_* against tests/golden/raytrace_{init,bumpy,C2}.npz
(outputs of the reference's own RayTracing.forward on the same SDF weights, rays, masks and injected fractions).
"""
import torch
import torch.nn.functional as F


# ---- utils/rend_util.py:141-162 ---------------------------------------------------------------------------------
def sphere_intersection(cam_loc, ray_dirs, r=1.0):
    """near/far ray parameters of the bounding sphere [B,P,2] (clamped at 0) and the hit mask [B,P]."""
    B, P, _ = ray_dirs.shape
    dot = torch.bmm(ray_dirs, cam_loc.unsqueeze(-1)).squeeze()
    disc = (dot ** 2 - (cam_loc.unsqueeze(-1).norm(2, 1) ** 2 - r ** 2)).reshape(-1)
    hit = disc > 0
    t = torch.zeros(B * P, 2)
    t[hit] = torch.sqrt(disc[hit]).unsqueeze(-1) * torch.tensor([-1.0, 1.0])
    t[hit] -= dot.reshape(-1)[hit].unsqueeze(-1)
    return t.reshape(B, P, 2).clamp_min(0.0), hit.reshape(B, P)


# ---- utils/rend_util.py:48-119 ----------------------------------------------------------------------------------
def quat_to_rot(q):
    q = F.normalize(q, dim=1)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y ** 2 + z ** 2), 2 * (y * x - z * w), 2 * (x * z + w * y),
            2 * (y * x + z * w), 1 - 2 * (x ** 2 + z ** 2), 2 * (y * z - x * w),
            2 * (z * x - y * w), 2 * (y * z + x * w), 1 - 2 * (x ** 2 + y ** 2)]
    return torch.stack(rows, -1).reshape(-1, 3, 3)


def camera_rays(uv, pose, intrinsics):
    """unit ray directions [B,P,3] and camera centres [B,3] from pixel coordinates (depth-1 lift, then world)."""
    if pose.shape[1] == 7:
        cam = pose[:, 4:]
        p = torch.eye(4).repeat(pose.shape[0], 1, 1)
        p[:, :3, :3] = quat_to_rot(pose[:, :4])
        p[:, :3, 3] = cam
    else:
        cam, p = pose[:, :3, 3], pose
    fx, fy = intrinsics[:, 0, 0].unsqueeze(-1), intrinsics[:, 1, 1].unsqueeze(-1)
    cx, cy = intrinsics[:, 0, 2].unsqueeze(-1), intrinsics[:, 1, 2].unsqueeze(-1)
    sk = intrinsics[:, 0, 1].unsqueeze(-1)
    x, y = uv[:, :, 0], uv[:, :, 1]
    z = torch.ones_like(x)
    xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    yl = (y - cy) / fy * z
    pix = torch.stack((xl, yl, z, torch.ones_like(z)), -1).permute(0, 2, 1)
    world = torch.bmm(p, pix).permute(0, 2, 1)[:, :, :3]
    return F.normalize(world - cam[:, None, :], dim=2), cam


# ---- model/sample_network.py:10-20 ------------------------------------------------------------------------------
def sample_point(sdf_theta, sdf0, grad0, t0, cam, dirs):
    """IDR eq. 3: first-order re-parametrisation of the hit point in the network parameters."""
    denom = torch.bmm(grad0.view(-1, 1, 3), dirs.detach().view(-1, 3, 1)).squeeze(-1)
    return cam + (t0 - (sdf_theta - sdf0) / denom) * dirs


# ---- model/ray_tracing.py --------------------------------------------------------------------------------------
class RayTraceRef:
    def __init__(self, object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=1,
                 sphere_tracing_iters=10, n_steps=100, n_secant_steps=8):
        self.r, self.thr = object_bounding_sphere, sdf_threshold
        self.ls_step, self.ls_iters = line_search_step, line_step_iters
        self.max_iters, self.n_steps, self.n_secant = sphere_tracing_iters, n_steps, n_secant_steps
        self.training = True
        self.steps = None         # optional injected U(0,1) fractions of the closest-approach search (:277)
        self.sdf_evals = 0

    def _sdf(self, sdf, pts):
        self.sdf_evals += pts.shape[0]
        return sdf(pts)

    # :26-95
    def __call__(self, sdf, cam_loc, object_mask, ray_dirs):
        B, P, _ = ray_dirs.shape
        N = B * P
        dirs = ray_dirs.reshape(N, 3)
        cams = cam_loc.unsqueeze(1).repeat(1, P, 1).reshape(N, 3)
        object_mask = object_mask.reshape(N)
        t_sph, hit = sphere_intersection(cam_loc, ray_dirs, self.r)
        t_sph, hit = t_sph.reshape(N, 2), hit.reshape(N)

        pts, open_near, t_near, t_far, t_lo, t_hi = self._march(sdf, cams, dirs, hit, t_sph)
        net = t_near < t_far
        if open_near.any():
            s_pts, s_net, s_t = self._sample(sdf, cams, dirs, object_mask, t_near, t_far, open_near)
            pts[open_near], t_near[open_near], net[open_near] = s_pts, s_t, s_net
        if not self.training:
            return pts, net, t_near

        inside_missed = ~net & object_mask & ~open_near          # P_in rays the network misses
        outside = ~object_mask & ~open_near                      # P_out rays
        loss_rays = inside_missed | outside
        no_sphere = loss_rays & ~hit
        if no_sphere.any():          # closest point of the ray to the origin (:77-82)
            c, d = cams[no_sphere], dirs[no_sphere]
            t_near[no_sphere] = -torch.bmm(d.view(-1, 1, 3), c.view(-1, 3, 1)).squeeze()
            pts[no_sphere] = c + t_near[no_sphere].unsqueeze(1) * d
        sel = loss_rays & hit
        if sel.any():                # minimal-SDF point between [t_lo, t_hi] (:84-92, 270-298)
            t_lo[net & outside] = t_near[net & outside]
            m_pts, m_t = self._closest(sdf, cams[sel], dirs[sel], t_lo[sel], t_hi[sel])
            pts[sel], t_near[sel] = m_pts, m_t
        return pts, net, t_near

    # :98-187
    def _march(self, sdf, cams, dirs, hit, t_sph):
        N = dirs.shape[0]
        at = lambda t: cams + t.unsqueeze(-1) * dirs                           # noqa: E731
        go_a, go_b = hit.clone(), hit.clone()                                  # near / far end still marching
        t_a, t_b = torch.zeros(N), torch.zeros(N)
        t_a[hit], t_b[hit] = t_sph[hit, 0], t_sph[hit, 1]
        p_a, p_b = torch.zeros(N, 3), torch.zeros(N, 3)
        p_a[hit], p_b[hit] = at(t_sph[:, 0])[hit], at(t_sph[:, 1])[hit]
        t_lo, t_hi = t_a.clone(), t_b.clone()

        def evaluate(points, which, into):
            if which.any():
                into[which] = self._sdf(sdf, points[which])

        nxt_a, nxt_b = torch.zeros(N), torch.zeros(N)
        evaluate(p_a, go_a, nxt_a)
        evaluate(p_b, go_b, nxt_b)
        for it in range(self.max_iters + 1):
            d_a = torch.where(go_a, nxt_a, torch.zeros(N))
            d_a[d_a <= self.thr] = 0
            d_b = torch.where(go_b, nxt_b, torch.zeros(N))
            d_b[d_b <= self.thr] = 0
            go_a &= d_a > self.thr
            go_b &= d_b > self.thr
            if it == self.max_iters or not (go_a.any() or go_b.any()):
                break
            t_a, t_b = t_a + d_a, t_b - d_b
            p_a, p_b = at(t_a), at(t_b)
            nxt_a, nxt_b = torch.zeros(N), torch.zeros(N)
            evaluate(p_a, go_a, nxt_a)
            evaluate(p_b, go_b, nxt_b)
            in_a, in_b = nxt_a < 0, nxt_b < 0                                   # stepped through the surface
            for k in range(self.ls_iters):
                if not (in_a.any() or in_b.any()):
                    break
                back = (1 - self.ls_step) / (2 ** k)
                t_a[in_a] -= back * d_a[in_a]
                p_a[in_a] = at(t_a)[in_a]
                t_b[in_b] += back * d_b[in_b]
                p_b[in_b] = at(t_b)[in_b]
                evaluate(p_a, in_a, nxt_a)
                evaluate(p_b, in_b, nxt_b)
                in_a, in_b = nxt_a < 0, nxt_b < 0
            go_a &= t_a < t_b
            go_b &= t_a < t_b
        return p_a, go_a, t_a, t_b, t_lo, t_hi

    # :189-249
    def _sample(self, sdf, cams, dirs, object_mask, t_near, t_far, which):
        n = self.n_steps
        c, d, lo, hi = cams[which], dirs[which], t_near[which], t_far[which]
        M = c.shape[0]
        ts = lo.unsqueeze(-1) + torch.linspace(0, 1, steps=n).view(1, n) * (hi - lo).unsqueeze(-1)
        pts = c.unsqueeze(1) + ts.unsqueeze(-1) * d.unsqueeze(1)
        vals = torch.cat([self._sdf(sdf, chunk) for chunk in torch.split(pts.reshape(-1, 3), 10000)]).reshape(M, n)
        # first sign change: argmin of sign * (n, n-1, ..., 1)
        first = torch.argmin(torch.sign(vals) * torch.arange(n, 0, -1).float().view(1, n), -1)
        row = torch.arange(M)
        out_pts, out_t = pts[row, first].clone(), ts[row, first].clone()
        neg = vals[row, first] < 0
        inside = object_mask[which]
        fallback = ~(inside & neg)
        if fallback.any():           # P_out pixels: the sample of minimal SDF instead
            j = torch.argmin(vals[fallback], -1)
            k = torch.arange(int(fallback.sum()))
            out_pts[fallback], out_t[fallback] = pts[fallback][k, j], ts[fallback][k, j]
        refine = (neg & inside) if self.training else neg
        if refine.any():
            f = first[refine]
            k = torch.arange(f.shape[0])
            z_hi, v_hi = ts[refine][k, f], vals[refine][k, f]
            z_lo, v_lo = ts[refine][k, f - 1], vals[refine][k, f - 1]          # f == 0 wraps around (negative index)
            z = self._secant(sdf, v_lo, v_hi, z_lo, z_hi, c[refine], d[refine])
            out_pts[refine], out_t[refine] = c[refine] + z.unsqueeze(-1) * d[refine], z
        return out_pts, neg, out_t

    # :251-268
    def _secant(self, sdf, v_lo, v_hi, z_lo, z_hi, c, d):
        v_lo, v_hi, z_lo, z_hi = v_lo.clone(), v_hi.clone(), z_lo.clone(), z_hi.clone()
        z = -v_lo * (z_hi - z_lo) / (v_hi - v_lo) + z_lo
        for _ in range(self.n_secant):
            v = self._sdf(sdf, c + z.unsqueeze(-1) * d)
            pos, neg = v > 0, v < 0
            z_lo[pos], v_lo[pos] = z[pos], v[pos]
            z_hi[neg], v_hi[neg] = z[neg], v[neg]
            z = -v_lo * (z_hi - z_lo) / (v_hi - v_lo) + z_lo
        return z

    # :270-298
    def _closest(self, sdf, c, d, lo, hi):
        n = self.n_steps
        M = c.shape[0]
        u = self.steps if self.steps is not None else torch.empty(n).uniform_(0.0, 1.0)
        ts = u.unsqueeze(0).repeat(M, 1) * (hi.unsqueeze(-1) - lo.unsqueeze(-1)) + lo.unsqueeze(-1)
        pts = c.unsqueeze(1).repeat(1, n, 1) + ts.unsqueeze(-1) * d.unsqueeze(1).repeat(1, n, 1)
        vals = torch.cat([self._sdf(sdf, chunk) for chunk in torch.split(pts.reshape(-1, 3), 10000)]).reshape(M, n)
        j = vals.argmin(-1)
        row = torch.arange(M)
        return pts[row, j], ts[row, j]
