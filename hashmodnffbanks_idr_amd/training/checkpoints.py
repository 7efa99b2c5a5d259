"""Checkpoint wire format of the reference runner (training/idr_train.py:140-165 load, :181-216 save).

    <root>/ModelParameters/{<epoch>,latest}.pth       {"epoch", "model_state_dict"}
    <root>/OptimizerParameters/{...}.pth              {"epoch", "optimizer_state_dict"}
    <root>/SchedulerParameters/{...}.pth              {"epoch", "scheduler_state_dict"}
    <root>/OptimizerCamParameters, <root>/CamParameters   (only with train_cameras)

so a run can be continued by either code base: the model's state_dict keys are the reference's
(`implicit_network.embed_model.embedder_obj.levels.{l}.embedding.weight`, `lin{l}.weight_g|weight_v|bias`, ...) and
training.optim.ClipAdam writes torch.optim.Adam's optimizer state.  Files are read with ``weights_only=True``
(tensors, dicts, numbers only - nothing from the file is executed).
"""
import os

import torch

MODEL_DIR, OPT_DIR, SCHED_DIR = "ModelParameters", "OptimizerParameters", "SchedulerParameters"
OPT_CAM_DIR, CAM_DIR = "OptimizerCamParameters", "CamParameters"


def _save_pair(root, sub, epoch, payload):
    d = os.path.join(root, sub)
    os.makedirs(d, exist_ok=True)
    torch.save(payload, os.path.join(d, str(epoch) + ".pth"))
    torch.save(payload, os.path.join(d, "latest.pth"))


def save_checkpoints(root, epoch, model, optimizer, scheduler=None, pose_vecs=None, optimizer_cam=None):
    _save_pair(root, MODEL_DIR, epoch, {"epoch": epoch, "model_state_dict": model.state_dict()})
    _save_pair(root, OPT_DIR, epoch, {"epoch": epoch, "optimizer_state_dict": optimizer.state_dict()})
    if scheduler is not None:
        _save_pair(root, SCHED_DIR, epoch, {"epoch": epoch, "scheduler_state_dict": scheduler.state_dict()})
    if pose_vecs is not None:
        _save_pair(root, OPT_CAM_DIR, epoch, {"epoch": epoch, "optimizer_cam_state_dict": optimizer_cam.state_dict()})
        _save_pair(root, CAM_DIR, epoch, {"epoch": epoch, "pose_vecs_state_dict": pose_vecs.state_dict()})


def _load(root, sub, checkpoint, map_location):
    return torch.load(os.path.join(root, sub, str(checkpoint) + ".pth"), map_location=map_location, weights_only=True)


def load_checkpoints(root, model, optimizer=None, scheduler=None, pose_vecs=None, optimizer_cam=None,
                     checkpoint="latest", map_location=None):
    """Returns the epoch stored with the model (the runner's ``start_epoch``)."""
    saved = _load(root, MODEL_DIR, checkpoint, map_location)
    model.load_state_dict(saved["model_state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(_load(root, OPT_DIR, checkpoint, map_location)["optimizer_state_dict"])
    if scheduler is not None:
        scheduler.load_state_dict(_load(root, SCHED_DIR, checkpoint, map_location)["scheduler_state_dict"])
    if pose_vecs is not None:
        optimizer_cam.load_state_dict(_load(root, OPT_CAM_DIR, checkpoint, map_location)["optimizer_cam_state_dict"])
        pose_vecs.load_state_dict(_load(root, CAM_DIR, checkpoint, map_location)["pose_vecs_state_dict"])
    return saved["epoch"]
