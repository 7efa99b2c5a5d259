"""DTU-style scene loader feeding the hot path (reference: code/datasets/scene_dataset.py:8-160,
DATA_CONVENTION.md): `<root>/<data_dir>/scan<id>/{image,mask}/*.png + cameras.npz` with per-image
`world_mat_i`, `scale_mat_i`; P_i = (world_mat_i @ scale_mat_i)[:3,:4] is decomposed into intrinsics and a
camera-to-world pose (utils/rend_util.load_K_Rt_from_P).  Samples are the dicts IDRNetwork.forward consumes
(`uv`, `intrinsics`, `pose`, `object_mask`) plus the `rgb` ground truth; `change_sampling_idx(n)` draws the
per-iteration pixel subset.  Same constructor and methods as the reference class; `root` replaces its
hard-coded '../data'.  Images are decoded with Pillow."""
import os

import numpy as np
import torch

from ..utils import general as utils
from ..utils import rend_util


class SceneDataset(torch.utils.data.Dataset):
    def __init__(self, train_cameras, data_dir, img_res, scan_id=0, cam_file=None, root='../data'):
        self.instance_dir = os.path.join(root, data_dir, 'scan{0}'.format(scan_id))
        if not os.path.exists(self.instance_dir):
            raise FileNotFoundError(f"Data directory is empty: {self.instance_dir}")
        self.img_res = img_res
        self.total_pixels = img_res[0] * img_res[1]
        self.sampling_idx = None
        self.train_cameras = train_cameras

        image_paths = sorted(utils.glob_imgs(os.path.join(self.instance_dir, 'image')))
        mask_paths = sorted(utils.glob_imgs(os.path.join(self.instance_dir, 'mask')))
        self.n_images = len(image_paths)
        self.cam_file = os.path.join(self.instance_dir, cam_file if cam_file is not None else 'cameras.npz')

        self.intrinsics_all, self.pose_all = [], []
        for intrinsics, pose in self._cameras(self.cam_file, scaled=True):
            self.intrinsics_all.append(torch.from_numpy(intrinsics).float())
            self.pose_all.append(torch.from_numpy(pose).float())
        # [H*W, 3] in [-1,1] and [H*W] bool, row-major pixels
        self.rgb_images = [torch.from_numpy(rend_util.load_rgb(p).reshape(3, -1).transpose(1, 0).copy()).float()
                           for p in image_paths]
        self.object_masks = [torch.from_numpy(rend_util.load_mask(p).reshape(-1)).bool() for p in mask_paths]
        # pixel coordinates (x, y) of every pixel, row-major: the reference flips np.mgrid's (row, col)
        rows, cols = np.mgrid[0:img_res[0], 0:img_res[1]].astype(np.int32)
        self._uv = torch.from_numpy(np.stack([cols, rows], 0).reshape(2, -1).transpose(1, 0).copy()).float()

    def _cameras(self, cam_file, scaled):
        camera_dict = np.load(cam_file)
        for idx in range(self.n_images):
            P = camera_dict['world_mat_%d' % idx].astype(np.float32)
            if scaled:
                P = P @ camera_dict['scale_mat_%d' % idx].astype(np.float32)
            yield rend_util.load_K_Rt_from_P(None, P[:3, :4])

    def __len__(self):
        return self.n_images

    def __getitem__(self, idx):
        sample = {"object_mask": self.object_masks[idx], "uv": self._uv, "intrinsics": self.intrinsics_all[idx]}
        ground_truth = {"rgb": self.rgb_images[idx]}
        if self.sampling_idx is not None:
            ground_truth["rgb"] = self.rgb_images[idx][self.sampling_idx, :]
            sample["object_mask"] = self.object_masks[idx][self.sampling_idx]
            sample["uv"] = self._uv[self.sampling_idx, :]
        if not self.train_cameras:
            sample["pose"] = self.pose_all[idx]
        return idx, sample, ground_truth

    def collate_fn(self, batch_list):
        """list of (idx, sample, ground_truth) -> (LongTensor idx, stacked sample dict, stacked gt dict)."""
        parsed = []
        for entry in zip(*batch_list):
            if isinstance(entry[0], dict):
                parsed.append({k: torch.stack([obj[k] for obj in entry]) for k in entry[0]})
            else:
                parsed.append(torch.LongTensor(entry))
        return tuple(parsed)

    def change_sampling_idx(self, sampling_size):
        self.sampling_idx = None if sampling_size == -1 else torch.randperm(self.total_pixels)[:sampling_size]

    def get_scale_mat(self):
        return np.load(self.cam_file)['scale_mat_0']

    def get_gt_pose(self, scaled=False):
        """ground-truth poses without (or with) the normalisation to the unit sphere -> [n_images, 4, 4]."""
        return torch.stack([torch.from_numpy(pose).float() for _, pose in self._cameras(self.cam_file, scaled)], 0)

    def get_pose_init(self):
        """noisy initial poses of the linear method as (quaternion, centre) [n_images, 7] (cameras_linear_init.npz)."""
        poses = torch.stack([torch.from_numpy(pose).float() for _, pose in
                             self._cameras(os.path.join(self.instance_dir, 'cameras_linear_init.npz'), True)], 0)
        return torch.cat([rend_util.rot_to_quat(poses[:, :3, :3]), poses[:, :3, 3]], 1)
