"""get_class: dotted-path plugin loader with the reference's semantics
(reference: code/utils/general.py:9-15) - the hook through which
``train.model_class = hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer.IDRNetwork``
drops this package into training/exp_runner.py."""
import importlib


def get_class(kls):
    parts = kls.split('.')
    module = importlib.import_module(".".join(parts[:-1]))
    return getattr(module, parts[-1])
