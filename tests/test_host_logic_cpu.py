"""CPU tests of host logic: seed-for-seed initialisation parity with the reference constructor,
camera helpers, state_dict key compatibility of the whole IDRNetwork."""
import numpy as np
import pytest
import torch

from helpers import idr_conf


@pytest.mark.parametrize("cfg", ["C1", "shipped"])
def test_init_rng_parity(golden, cfg):
    """torch.manual_seed(s); IDRNetwork(conf) produces the reference's freshly initialised parameters."""
    from hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer import IDRNetwork
    g = golden("init_rng")
    torch.manual_seed(1234)
    model = IDRNetwork(idr_conf(cfg))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    names = [str(n) for n in g[f"{cfg}:names"]]
    assert sorted(sd.keys()) == sorted(names)
    for k in names:
        a = sd[k].numpy().astype(np.float64)
        assert np.array_equal(sd[k].numpy().reshape(-1)[:8], g[f"{cfg}:{k}:head"]), k
        assert abs(a.sum() - float(g[f"{cfg}:{k}:sum"])) <= 1e-9 * max(1.0, abs(float(g[f'{cfg}:{k}:abs']))), k
        assert abs(np.abs(a).sum() - float(g[f"{cfg}:{k}:abs"])) <= 1e-9 * max(1.0, float(g[f"{cfg}:{k}:abs"])), k


def test_camera_params(golden):
    from hashmodnffbanks_idr_amd.utils import rend_util
    g = golden("camera")
    T = torch.from_numpy
    d7, c7 = rend_util.get_camera_params(T(g["uv"]), T(g["pose7"]), T(g["K"]))
    np.testing.assert_allclose(d7.numpy(), g["dirs7"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(c7.numpy(), g["cam7"], rtol=0, atol=0)
    d4, c4 = rend_util.get_camera_params(T(g["uv"]), T(g["pose44"]), T(g["K"]))
    np.testing.assert_allclose(d4.numpy(), g["dirs44"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rend_util.quat_to_rot(T(g["pose7"][:, :4])).numpy(), g["R"], rtol=1e-6, atol=1e-6)


def test_get_class_plugin_point():
    from hashmodnffbanks_idr_amd.utils.general import get_class
    cls = get_class("hashmodnffbanks_idr_amd.model.implicit_differentiable_renderer.IDRNetwork")
    assert cls.__name__ == "IDRNetwork"
    assert get_class("hashmodnffbanks_idr_amd.model.loss.IDRLoss").__name__ == "IDRLoss"


def test_pack_layer_layout():
    """w_packed[((u*n_oct+g)*64+l)*4+s] == W[32u+(l&31)][8g+4(l>>5)+s] (include/hashmod.h contract)."""
    from hashmodnffbanks_idr_amd import ops
    W = torch.arange(45 * 83, dtype=torch.float32).reshape(45, 83)
    Wp, n_tiles, octs, srcs = ops.pack_mlp_layer(W, [(0, 64), (1, 19)])
    assert n_tiles == 2 and octs == [8, 3] and srcs == [0, 1]
    flat = Wp.reshape(-1)
    n_oct = 11
    padded = torch.zeros(64, 88)
    padded[:45, :64] = W[:, :64]
    padded[:45, 64:83] = W[:, 64:]
    for (u, g, l, s) in [(0, 0, 0, 0), (1, 10, 63, 3), (0, 8, 33, 2), (1, 3, 12, 1), (1, 7, 44, 0)]:
        assert flat[((u * n_oct + g) * 64 + l) * 4 + s] == padded[32 * u + (l & 31), 8 * g + 4 * (l >> 5) + s]
