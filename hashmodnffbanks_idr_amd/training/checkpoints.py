"""Checkpoint wire format of the reference runner (training/idr_train.py:140-165 load, :181-216 save).

    <root>/ModelParameters/{<epoch>,latest}.pth       {"epoch", "model_state_dict"}
    <root>/OptimizerParameters/{...}.pth              {"epoch", "optimizer_state_dict"}
    <root>/SchedulerParameters/{...}.pth              {"epoch", "scheduler_state_dict"}
    <root>/OptimizerCamParameters, <root>/CamParameters   (only with train_cameras)

so a run can be continued by either code base: the model's state_dict keys are the reference's
(`implicit_network.embed_model.embedder_obj.levels.{l}.embedding.weight`, `lin{l}.weight_g|weight_v|bias`, ...) and
training.optim.ClipAdam writes torch.optim.Adam's optimizer state.

Optimizer state needs one translation at this boundary: here a hash grid is ONE fused `table` parameter, in the
reference it is L parameters `levels.{l}.embedding.weight` (same position in `model.parameters()`, checked against the
reference's own parameter list in tests/test_host_logic_cpu.py).  `optimizer_state_to_reference` splits the table's
`exp_avg` / `exp_avg_sq` (and repeats its `step`) into the L per-level entries and renumbers the parameter indices;
`optimizer_state_from_reference` fuses them again.  save_checkpoints / load_checkpoints apply them, so
OptimizerParameters/*.pth has the reference's layout.  Files are read with ``weights_only=True`` (tensors, dicts,
numbers only - nothing from the file is executed).
"""
import os

import torch


def _reference_layout(model):
    """[(own parameter index, None | (row0, row1))] in the reference's parameter order."""
    from ..model.embeddings.hashGridEmbedding import MultiResHashGridMLP
    tables = {}
    for mod in model.modules():
        if isinstance(mod, MultiResHashGridMLP):
            off = [int(v) for v in mod.desc.row_off]
            tables[id(mod.table)] = [(off[l], off[l + 1]) for l in range(mod.n_levels)]
    layout = []
    for k, p in enumerate(model.parameters()):
        if id(p) in tables:
            layout += [(k, rows) for rows in tables[id(p)]]
        else:
            layout.append((k, None))
    return layout


def optimizer_state_to_reference(model, sd):
    """state_dict of an Adam-like optimizer over model.parameters() -> the same state over the reference's parameters."""
    if len(sd["param_groups"]) != 1:
        raise ValueError("one parameter group expected (the reference runner's optimizer, idr_train.py:128)")
    layout = _reference_layout(model)
    state = {}
    for j, (k, rows) in enumerate(layout):
        st = sd["state"].get(k)
        if st is None:
            continue
        state[j] = {name: (v[rows[0]:rows[1]].clone() if (rows is not None and torch.is_tensor(v) and v.dim() > 0) else
                           (v.clone() if torch.is_tensor(v) else v)) for name, v in st.items()}
    # every key torch.optim.Adam.step() reads must be in the group, whichever Adam-like optimizer wrote `sd`
    group = dict(_ADAM_GROUP_DEFAULTS)
    group.update(sd["param_groups"][0])
    group["params"] = list(range(len(layout)))
    return {"state": state, "param_groups": [group]}


_ADAM_GROUP_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                            differentiable=False, fused=None, decoupled_weight_decay=False)


def optimizer_state_from_reference(model, sd):
    """inverse of optimizer_state_to_reference: per-level entries of a hash grid are fused into its table's state."""
    if len(sd["param_groups"]) != 1:
        raise ValueError("one parameter group expected (the reference runner's optimizer, idr_train.py:128)")
    layout = _reference_layout(model)
    if len(sd["param_groups"][0]["params"]) != len(layout):
        raise ValueError(f"optimizer checkpoint covers {len(sd['param_groups'][0]['params'])} parameters, "
                         f"the model has {len(layout)} in the reference's layout")
    by_own = {}
    for j, (k, rows) in enumerate(layout):
        by_own.setdefault(k, []).append((j, rows))
    state = {}
    for k, parts in by_own.items():
        sts = [sd["state"].get(j) for j, _ in parts]
        if all(st is None for st in sts):
            continue
        if any(st is None for st in sts):
            raise ValueError("optimizer checkpoint has state for some levels of a hash grid only")
        if parts[0][1] is None:
            state[k] = {name: (v.clone() if torch.is_tensor(v) else v) for name, v in sts[0].items()}
            continue
        fused = {}
        for name, v in sts[0].items():
            if torch.is_tensor(v) and v.dim() > 0:
                fused[name] = torch.cat([st[name] for st in sts], 0)
            else:
                if any(float(st[name]) != float(v) for st in sts):
                    raise ValueError(f"levels of one hash grid disagree on '{name}'")
                fused[name] = v.clone() if torch.is_tensor(v) else v
        state[k] = fused
    group = dict(sd["param_groups"][0])
    group["params"] = list(range(len(list(model.parameters()))))
    return {"state": state, "param_groups": [group]}

MODEL_DIR, OPT_DIR, SCHED_DIR = "ModelParameters", "OptimizerParameters", "SchedulerParameters"
OPT_CAM_DIR, CAM_DIR = "OptimizerCamParameters", "CamParameters"


def _save_pair(root, sub, epoch, payload):
    d = os.path.join(root, sub)
    os.makedirs(d, exist_ok=True)
    torch.save(payload, os.path.join(d, str(epoch) + ".pth"))
    torch.save(payload, os.path.join(d, "latest.pth"))


def save_checkpoints(root, epoch, model, optimizer, scheduler=None, pose_vecs=None, optimizer_cam=None):
    _save_pair(root, MODEL_DIR, epoch, {"epoch": epoch, "model_state_dict": model.state_dict()})
    _save_pair(root, OPT_DIR, epoch, {"epoch": epoch, "optimizer_state_dict":
                                      optimizer_state_to_reference(model, optimizer.state_dict())})
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
        optimizer.load_state_dict(optimizer_state_from_reference(
            model, _load(root, OPT_DIR, checkpoint, map_location)["optimizer_state_dict"]))
    if scheduler is not None:
        scheduler.load_state_dict(_load(root, SCHED_DIR, checkpoint, map_location)["scheduler_state_dict"])
    if pose_vecs is not None:
        optimizer_cam.load_state_dict(_load(root, OPT_CAM_DIR, checkpoint, map_location)["optimizer_cam_state_dict"])
        pose_vecs.load_state_dict(_load(root, CAM_DIR, checkpoint, map_location)["pose_vecs_state_dict"])
    return saved["epoch"]
