"""Phase timing of ONE 64-point tile of the fused SDF kernel (csrc/hm_sdf.hip, sdf_fwd_kernel): cycle stamps taken by
workgroup 0 / thread 0 on its second tile at the encode, and per layer at [end of the MFMA k-loop, after the first
barrier, end of the epilogue, after the second barrier].  Needs a probe build of the library:

    python scripts/sdf_phase_probe.py --build     (here: compiles csrc/*.hip, hm_sdf.hip with -DHM_SDF_PHASE_PROBE, into
                                                   scripts/libhashmod_probe.so)
    HM_LIB_PATH=scripts/libhashmod_probe.so python scripts/sdf_phase_probe.py      (on the GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/tests", R + "/tests/golden"]
PROBE = os.path.join(R, "scripts", "libhashmod_probe.so")

if "--build" in sys.argv:
    from hashmodnffbanks_idr_amd import build as B
    objs = []
    for src in B.SOURCES:
        obj = os.path.join("/tmp", "probe_" + src.replace(".hip", ".o"))
        cmd = [B.HIPCC] + B.FLAGS + (["-DHM_SDF_PHASE_PROBE=1"] if src == "hm_sdf.hip" else []) + \
              ["-c", os.path.join(B.CSRC, src), "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE] + objs)
    print(PROBE)
    sys.exit(0)

import numpy as np
import torch
import bench
from hashmodnffbanks_idr_amd import _lib

model = bench._build("C2", torch.device("cuda", 0), 0.0)
net = model.implicit_network
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand((204800, 3), generator=g) * 2 - 1).cuda()
net.sdf_tile_points = 64
for _ in range(3):
    net.sdf(x)
torch.cuda.synchronize()
ts = (C.c_ulonglong * 128)()
fn = _lib.lib().hm_probe_read
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(ts, 128) == 0
t = np.asarray(list(ts), dtype=np.int64)
MHZ = 100.0   # wall_clock64() = s_memrealtime: 100 MHz constant clock
print("tile start -> encode done: %.2f us" % ((t[1] - t[0]) / MHZ))
tot_m = tot_b1 = tot_e = tot_b2 = 0.0
prev = t[1]
for li in range(8):
    m, b1, e, b2 = t[2 + 4 * li], t[3 + 4 * li], t[4 + 4 * li], t[5 + 4 * li]
    print("layer %d: k-loop %.2f us  barrier %.2f  epilogue %.2f  barrier %.2f" %
          (li, (m - prev) / MHZ, (b1 - m) / MHZ, (e - b1) / MHZ, (b2 - e) / MHZ))
    tot_m += m - prev; tot_b1 += b1 - m; tot_e += e - b1; tot_b2 += b2 - e
    prev = b2
print("last layer (VALU dot) + output: %.2f us" % ((t[100] - prev) / MHZ if t[100] > prev else float("nan")))
base = t[5 + 4 * 1]     # end of layer 1 (after its second barrier) = start of layer 2's k-loop for thread 0
print("layer 2, end of the k-loop per wave (us after layer 2 started): " +
      " ".join("w%d %.2f" % (w, (t[64 + w] - base) / MHZ) for w in range(8)))
print("shader clock over the tile: %.3f GHz (clock64 / wall_clock64 at 100 MHz)" %
      ((t[121] - t[120]) / ((t[100] - t[0]) / MHZ) / 1e3))
print("sums: k-loops %.1f  first barriers %.1f  epilogues %.1f  second barriers %.1f  (us)" %
      (tot_m / MHZ, tot_b1 / MHZ, tot_e / MHZ, tot_b2 / MHZ))
