"""Phase stamps of ONE 4-point tile of the small-tile SDF body (probe build: scripts/sdf_phase_probe.py --build;
run with HM_LIB_PATH=scripts/libhashmod_probe.so): where do the ~40 us of a small launch go that do not depend on the
size of the weights?  Per layer: [k-loop done, k quarters added, after the first barrier, after epilogue + second barrier]."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
import numpy as np, torch, bench
from hashmodnffbanks_idr_amd import _lib
model = bench._build("C2", torch.device("cuda", 0), 0.0)
net = model.implicit_network
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand((256, 3), generator=g) * 2 - 1).cuda()
net.sdf_tile_points = 4
for _ in range(3):
    net.sdf(x)
torch.cuda.synchronize()
ts = (C.c_ulonglong * 128)()
fn = _lib.lib().hm_probe_read
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(ts, 128) == 0
t = np.asarray(list(ts), dtype=np.int64)
M = 100.0
print("tile start -> encode done: %.2f us" % ((t[1] - t[0]) / M))
prev = t[1]
for li in range(8):
    a, b, c, d = t[2 + 4 * li], t[3 + 4 * li], t[4 + 4 * li], t[5 + 4 * li]
    print("layer %d: k-loop %.2f  quarter sums %.2f  barrier %.2f  epilogue + barrier %.2f" %
          (li, (a - prev) / M, (b - a) / M, (c - b) / M, (d - c) / M))
    prev = d
print("last layer + output: %.2f us;  whole tile %.2f us" % ((t[100] - prev) / M, (t[100] - t[0]) / M))
