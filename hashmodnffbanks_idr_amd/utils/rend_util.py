"""Camera / ray geometry helpers on the hot path (reference: code/utils/rend_util.py:48-162).
Image IO helpers of the reference (imageio / skimage / cv2) are out of scope.  All results are
elementwise fp32 torch expressions evaluated on the inputs' device (the reference hard-codes
``.cuda()``)."""
import torch
from torch.nn import functional as F


def quat_to_rot(q):
    """unit quaternion (w, x, y, z) -> rotation matrix (rend_util.py:102-119)."""
    q = F.normalize(q, dim=1)
    qr, qi, qj, qk = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.ones((q.shape[0], 3, 3), device=q.device, dtype=q.dtype)
    R[:, 0, 0] = 1 - 2 * (qj ** 2 + qk ** 2)
    R[:, 0, 1] = 2 * (qj * qi - qk * qr)
    R[:, 0, 2] = 2 * (qi * qk + qr * qj)
    R[:, 1, 0] = 2 * (qj * qi + qk * qr)
    R[:, 1, 1] = 1 - 2 * (qi ** 2 + qk ** 2)
    R[:, 1, 2] = 2 * (qj * qk - qi * qr)
    R[:, 2, 0] = 2 * (qk * qi - qj * qr)
    R[:, 2, 1] = 2 * (qj * qk + qi * qr)
    R[:, 2, 2] = 1 - 2 * (qi ** 2 + qj ** 2)
    return R


def lift(x, y, z, intrinsics):
    """pixel (x, y) at depth z -> homogeneous camera-space point (rend_util.py:86-100)."""
    intrinsics = intrinsics.to(x.device)
    fx = intrinsics[:, 0, 0].unsqueeze(-1)
    fy = intrinsics[:, 1, 1].unsqueeze(-1)
    cx = intrinsics[:, 0, 2].unsqueeze(-1)
    cy = intrinsics[:, 1, 2].unsqueeze(-1)
    sk = intrinsics[:, 0, 1].unsqueeze(-1)
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    return torch.stack((x_lift, y_lift, z, torch.ones_like(z)), dim=-1)


def get_camera_params(uv, pose, intrinsics):
    """uv [B,N,2], pose [B,4,4] or [B,7] (quaternion + centre), K [B,4,4] -> unit ray directions
    [B,N,3] and camera centres [B,3] (rend_util.py:48-75)."""
    if pose.shape[1] == 7:
        cam_loc = pose[:, 4:]
        R = quat_to_rot(pose[:, :4])
        p = torch.eye(4, device=pose.device).repeat(pose.shape[0], 1, 1).float()
        p[:, :3, :3] = R
        p[:, :3, 3] = cam_loc
    else:
        cam_loc = pose[:, :3, 3]
        p = pose
    batch_size, num_samples, _ = uv.shape
    depth = torch.ones((batch_size, num_samples), device=uv.device)
    x_cam = uv[:, :, 0].view(batch_size, -1)
    y_cam = uv[:, :, 1].view(batch_size, -1)
    z_cam = depth.view(batch_size, -1)
    pixel_points_cam = lift(x_cam, y_cam, z_cam, intrinsics=intrinsics).permute(0, 2, 1)
    world_coords = torch.bmm(p, pixel_points_cam).permute(0, 2, 1)[:, :, :3]
    ray_dirs = F.normalize(world_coords - cam_loc[:, None, :], dim=2)
    return ray_dirs, cam_loc


def get_sphere_intersection(cam_loc, ray_directions, r=1.0):
    """near/far ray parameters of the bounding sphere |p| = r, clamped at 0, and the hit mask
    (rend_util.py:141-162).  cam_loc [B,3], ray_directions [B,N,3] -> [B,N,2], [B,N].
    Written with torch.where instead of the reference's boolean-mask assignment: same values, static
    shapes, no device->host synchronisation (so it can sit inside a captured HIP graph)."""
    n_imgs, n_pix, _ = ray_directions.shape
    c = cam_loc.unsqueeze(-1)
    ray_cam_dot = torch.bmm(ray_directions, c).reshape(-1)
    under_sqrt = (ray_cam_dot ** 2 - (c.norm(2, 1) ** 2 - r ** 2).repeat_interleave(n_pix, 0).reshape(-1))
    mask_intersect = under_sqrt > 0
    root = torch.sqrt(torch.where(mask_intersect, under_sqrt, torch.ones_like(under_sqrt)))
    t = torch.stack([root * -1.0, root], -1) - ray_cam_dot.unsqueeze(-1)   # sqrt * [-1, 1] - <d, c>
    t = torch.where(mask_intersect.unsqueeze(-1), t, torch.zeros_like(t))
    t = t.reshape(n_imgs, n_pix, 2).clamp_min(0.0)
    return t, mask_intersect.reshape(n_imgs, n_pix)
