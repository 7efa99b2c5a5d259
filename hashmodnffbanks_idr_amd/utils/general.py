"""get_class: dotted-path plugin loader with the reference's semantics
(reference: code/utils/general.py:9-15) - the hook through which
``train.model_class = hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer.IDRNetwork``
drops this package into training/exp_runner.py."""
import importlib


def get_class(kls):
    parts = kls.split('.')
    module = importlib.import_module(".".join(parts[:-1]))
    return getattr(module, parts[-1])


def glob_imgs(path):
    """image files of a directory (general.py:17-21)."""
    import os
    from glob import glob
    imgs = []
    for ext in ['*.png', '*.jpg', '*.JPEG', '*.JPG']:
        imgs.extend(glob(os.path.join(path, ext)))
    return imgs


def split_input(model_input, total_pixels, n_pixels=10000):
    """full-image evaluation in pixel chunks (general.py:23-36): one shallow copy of the input dict per chunk with
    `uv` and `object_mask` narrowed to it; indices live on the inputs' device."""
    import torch
    order = torch.arange(total_pixels, device=model_input['uv'].device)
    chunks = []
    for idx in order.split(n_pixels):
        part = dict(model_input)
        part['uv'] = model_input['uv'].index_select(1, idx)
        part['object_mask'] = model_input['object_mask'].index_select(1, idx)
        chunks.append(part)
    return chunks


def merge_output(res, total_pixels, batch_size):
    """concatenate the per-chunk output dicts of split_input along the pixel axis (general.py:38-52)."""
    import torch
    merged = {}
    for key, first in res[0].items():
        if first is None:
            continue
        flat = first.dim() == 1
        width = 1 if flat else first.shape[-1]
        joined = torch.cat([r[key].reshape(batch_size, -1, width) for r in res], dim=1)
        merged[key] = joined.reshape(batch_size * total_pixels) if flat else joined.reshape(batch_size * total_pixels, -1)
    return merged
