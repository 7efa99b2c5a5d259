"""Camera / ray geometry helpers on the hot path (reference: code/utils/rend_util.py:48-162) and the data
front-end helpers next to it (rend_util.py:8-46: image / mask loading, projection-matrix decomposition) written
against Pillow / numpy / scipy, which this image has, instead of imageio / skimage / cv2, which it has not.
All ray-side results are elementwise fp32 torch expressions evaluated on the inputs' device (the reference
hard-codes ``.cuda()``)."""
import numpy as np
import torch
from torch.nn import functional as F


def load_rgb(path):
    """image file -> float32 [3, H, W] in [-1, 1] (rend_util.py:8-16: img_as_float32, then *2-1, CHW)."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    img = (img - 0.5) * 2.0
    return img.transpose(2, 0, 1)


def load_mask(path):
    """mask file -> bool [H, W]: grey level > 127.5 (rend_util.py:18-23; imageio's mode='F' is Pillow's 'F')."""
    from PIL import Image
    alpha = np.asarray(Image.open(path).convert("F"), dtype=np.float32)
    return alpha > 127.5


def decompose_projection(P):
    """P [3,4] = K [R | -R C]  ->  K (upper triangular, positive diagonal), R (rotation), C (camera centre).
    The RQ factorisation cv2.decomposeProjectionMatrix performs (rend_util.py:33), via scipy.linalg.rq with the
    signs fixed so that diag(K) > 0.  (cv2 is not available in this image, so this function is pinned by
    construct-and-recover tests rather than by cv2 outputs: parity with cv2 itself is unpinned.)"""
    from scipy.linalg import rq
    P = np.asarray(P, dtype=np.float64)
    M = P[:, :3]
    K, R = rq(M)
    D = np.diag(np.where(np.diag(K) < 0, -1.0, 1.0))
    K, R = K @ D, D @ R
    C = -np.linalg.solve(M, P[:, 3])
    return K, R, C


def load_K_Rt_from_P(filename, P=None):
    """(intrinsics [4,4], pose [4,4] camera-to-world) from a 3x4 projection matrix or a text file holding one
    (rend_util.py:25-46): intrinsics = K / K[2,2] in the upper-left 3x3, pose = [R^T | C]."""
    if P is None:
        lines = open(filename).read().splitlines()
        if len(lines) == 4:
            lines = lines[1:]
        lines = [[x[0], x[1], x[2], x[3]] for x in (x.split(" ") for x in lines)]
        P = np.asarray(lines).astype(np.float32).squeeze()
    K, R, C = decompose_projection(P)
    K = K / K[2, 2]
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.transpose()
    pose[:3, 3] = C
    return intrinsics, pose


def rot_to_quat(R):
    """rotation matrices [B,3,3] -> quaternions (w, x, y, z) [B,4] (rend_util.py:121-139; w = sqrt(1+trace)/2)."""
    w = torch.sqrt(1.0 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]) / 2
    d = 4 * w
    return torch.stack([w, (R[:, 2, 1] - R[:, 1, 2]) / d, (R[:, 0, 2] - R[:, 2, 0]) / d,
                        (R[:, 1, 0] - R[:, 0, 1]) / d], dim=1)


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
