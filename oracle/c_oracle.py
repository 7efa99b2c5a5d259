"""ctypes bindings of oracle/hm_oracle.c (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhm_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "hm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhm_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.hmo_level_table.argtypes = [C.c_int] * 5 + [_i32p, _u32p, _u64p]
        L.hmo_level_table.restype = C.c_int
        L.hmo_corner_ids.argtypes = [_f32p, C.c_int64, C.c_int32, C.c_uint32, _i32p, _u32p]
        L.hmo_corner_ids.restype = None
        enc = [C.c_int, C.c_int, _i32p, _u32p, _u64p, _f32p, C.c_int64, _f32p, _f32p, _f32p, C.c_int]
        L.hmo_encode_fwd.argtypes = enc
        L.hmo_encode_fwd.restype = None
        L.hmo_encode_bwd_table.argtypes = [C.c_int, C.c_int, _i32p, _u32p, _u64p, _f32p, C.c_int64, _f32p, _f32p,
                                           C.c_int]
        L.hmo_encode_bwd_table.restype = None
        L.hmo_fold_weight_norm.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.hmo_fold_weight_norm.restype = None
        L.hmo_mlp_fwd.argtypes = [C.c_int, _i32p, _i32p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int,
                                  _f32p, C.c_int, C.c_int64, C.c_float, C.c_int, _f32p]
        L.hmo_mlp_fwd.restype = C.c_int
        L.hmo_sphere_intersection.argtypes = [_f32p, _f32p, C.c_int64, C.c_float, _f32p, _u8p]
        L.hmo_sphere_intersection.restype = None
        L.hmo_num_threads.restype = C.c_int
        _lib = L
    return _lib


class Grid:
    """Level table of one multi-resolution hash grid (hashGridEmbedding.py:126-132)."""

    def __init__(self, n_levels, log2_hashmap_size, base_resolution, desired_resolution, F=2):
        self.L, self.F = int(n_levels), int(F)
        self.res = np.zeros(self.L, np.int32)
        self.rows = np.zeros(self.L, np.uint32)
        self.row_off = np.zeros(self.L + 1, np.uint64)
        rc = lib().hmo_level_table(self.L, log2_hashmap_size, base_resolution, desired_resolution, 3,
                                   self.res, self.rows, self.row_off)
        if rc != 0:
            raise ValueError("bad level table arguments")
        self.total_rows = int(self.row_off[-1])
        self.E = 3 + 2 * self.L + self.L * self.F


def corner_ids(x, res, rows):
    x = np.ascontiguousarray(x, np.float32)
    n = x.shape[0]
    xi = np.zeros((n, 3), np.int32)
    ids = np.zeros((n, 8), np.uint32)
    lib().hmo_corner_ids(x, n, int(res), int(rows), xi, ids)
    return xi, ids


def encode_fwd(grid, x, table, B, frac_mode=0):
    x = np.ascontiguousarray(x, np.float32)
    table = np.ascontiguousarray(table, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    assert table.shape == (grid.total_rows, grid.F) and B.shape == (3, grid.L)
    out = np.zeros((x.shape[0], grid.E), np.float32)
    lib().hmo_encode_fwd(grid.L, grid.F, grid.res, grid.rows, grid.row_off, x, x.shape[0], table, B, out, frac_mode)
    return out


def encode_bwd_table(grid, x, d_out, frac_mode=0):
    x = np.ascontiguousarray(x, np.float32)
    d_out = np.ascontiguousarray(d_out, np.float32)
    assert d_out.shape == (x.shape[0], grid.E)
    d_table = np.zeros((grid.total_rows, grid.F), np.float32)
    lib().hmo_encode_bwd_table(grid.L, grid.F, grid.res, grid.rows, grid.row_off, x, x.shape[0], d_out, d_table,
                               frac_mode)
    return d_table


def fold_weight_norm(v, g):
    v = np.ascontiguousarray(v, np.float32)
    g = np.ascontiguousarray(g, np.float32).reshape(-1)
    w = np.zeros_like(v)
    lib().hmo_fold_weight_norm(v, g, v.shape[0], v.shape[1], w)
    return w


def mlp_fwd(weights, biases, emb, skip_layer=4, beta=0.9 + 1e-4, apply_clamp=True):
    """weights: folded [out,in] per layer."""
    n_layers = len(weights)
    Ws = [np.ascontiguousarray(w, np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, np.float32) for b in biases]
    emb = np.ascontiguousarray(emb, np.float32)
    in_dims = np.asarray([w.shape[1] for w in Ws], np.int32)
    out_dims = np.asarray([w.shape[0] for w in Ws], np.int32)
    Wp = (C.c_void_p * n_layers)(*[w.ctypes.data for w in Ws])
    bp = (C.c_void_p * n_layers)(*[b.ctypes.data for b in bs])
    out = np.zeros((emb.shape[0], int(out_dims[-1])), np.float32)
    rc = lib().hmo_mlp_fwd(n_layers, in_dims, out_dims, Wp, bp, int(skip_layer), emb, emb.shape[1], emb.shape[0],
                           np.float32(beta), int(apply_clamp), out)
    if rc != 0:
        raise ValueError("layer dimension mismatch")
    return out


def sphere_intersection(cam, dirs, r=1.0):
    cam = np.ascontiguousarray(cam, np.float32).reshape(3)
    dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    t2 = np.zeros((dirs.shape[0], 2), np.float32)
    m = np.zeros(dirs.shape[0], np.uint8)
    lib().hmo_sphere_intersection(cam, dirs, dirs.shape[0], np.float32(r), t2, m)
    return t2, m.astype(bool)


class SdfOracle:
    """encode + folded MLP forward: the `implicit_network(x)` of the reference, on the CPU."""

    def __init__(self, grid, table, B, params, skip_layer=4, beta_param=0.9, frac_mode=0):
        self.grid, self.table, self.B = grid, table, B
        n = len([k for k in params if k.endswith("weight_v")])
        self.W = [fold_weight_norm(params[f"lin{l}.weight_v"], params[f"lin{l}.weight_g"]) for l in range(n)]
        self.b = [params[f"lin{l}.bias"] for l in range(n)]
        self.skip_layer = skip_layer
        self.beta = np.float32(abs(np.float32(beta_param))) + np.float32(1e-4)
        self.frac_mode = frac_mode

    def __call__(self, x):
        emb = encode_fwd(self.grid, x, self.table, self.B, self.frac_mode)
        return mlp_fwd(self.W, self.b, emb, self.skip_layer, self.beta, True)

    def sdf(self, x):
        return self(x)[:, 0]
