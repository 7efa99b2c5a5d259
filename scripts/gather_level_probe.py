"""Which levels cost the z-ordered gather its time?  Probe builds of the library whose gather kernel skips all levels
outside [lmin, lmax] (hm_encode.hip with -DHM_ENC_DIAG_LMIN / -DHM_ENC_DIAG_LMAX; wrong results, timing only):

    python scripts/gather_level_probe.py --build     (here: scripts/libhashmod_lv_<lmin>_<lmax>.so, git-ignored)
    python scripts/gather_level_probe.py             (on the GPU box: bench.py --only gather with each of them)
"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANGES = [(0, 15), (0, 10), (11, 15), (0, 5), (6, 10), (99, 99)]
# (lmin, lmax, extra define): no levels and no Fourier columns / no row stores / neither
EXTRA = [(99, 99, "HM_ENC_DIAG_NOFOURIER"), (99, 99, "HM_ENC_DIAG_NOSTORE"), (0, 15, "HM_ENC_DIAG_NOFOURIER"), (0, 15, "HM_ENC_DIAG_NOSTORE")]

if "--build" in sys.argv:
    sys.path.insert(0, R)
    from hashmodnffbanks_idr_amd import build as B
    for lo, hi, *ex in [r + () for r in RANGES] + EXTRA:
        objs = []
        for src in B.SOURCES:
            obj = os.path.join(B.CSRC, src.replace(".hip", ".o"))          # the product build's objects
            if src == "hm_encode.hip":
                obj = f"/tmp/probe_enc_{lo}_{hi}{''.join(ex)}.o"
                subprocess.check_call([B.HIPCC] + B.FLAGS + [f"-DHM_ENC_DIAG_LMIN={lo}", f"-DHM_ENC_DIAG_LMAX={hi}"] +
                                      [f"-D{e}=1" for e in ex] + ["-c",
                                                             os.path.join(B.CSRC, src), "-o", obj])
            objs.append(obj)
        out = os.path.join(R, "scripts", f"libhashmod_lv_{lo}_{hi}{''.join(ex)}.so")
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
        print(out)
    sys.exit(0)

for lo, hi, *ex in [r + () for r in RANGES] + EXTRA:
    env = dict(os.environ, HM_LIB_PATH=os.path.join(R, "scripts", f"libhashmod_lv_{lo}_{hi}{''.join(ex)}.so"))
    for cfg in ("C2",):
        out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--only", "gather", "--cfg", cfg], env=env,
                             capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        print(f"levels {lo:2d}..{hi:2d} {' '.join(ex):22s} {cfg}: {d['avg_launch_ms']:.4f} ms per launch (min {d['min_launch_ms']:.4f})", flush=True)
