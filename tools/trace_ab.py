"""GPU-box experiment: stage timings of the wavefront pipeline variants on a reduced C3
workload (same scene, 960x540, 64 spp). Usage: python tools/trace_ab.py [flags ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes

if os.environ.get("YART_LIB"):          # experiment variant built by tools/build_variant.sh
    api.LIB_PATH = os.path.join(ROOT, "yart_amd", "_variants", os.environ["YART_LIB"] + ".so")
    print("variant", os.environ["YART_LIB"], flush=True)
w, h, spp = (int(x) for x in os.environ.get("SIZE", "960x540x64").split("x"))
TEX, SKY = int(os.environ.get("TEX", 256)), int(os.environ.get("SKY", 256))
if os.environ.get("SCENE", "sponza") == "mclaren":        # BASELINE configs[4] scene (1.05 M triangles at DETAIL=1)
    scene, p = scenes.mclaren_class(w, h, spp, 8, detail=float(os.environ.get("DETAIL", 1.0)), tex=TEX, sky=SKY)
else:
    scene, p = scenes.sponza_class(w, h, spp, 8, tex=TEX, sky=SKY)
ds = api.DeviceScene(scene, device=0)
flags = [int(x) for x in sys.argv[1:]] or [0, 2]
ref = None
for rep in range(2):
    for f in flags:
        img, st = ds.render(p, flags=f)
        if ref is None:
            ref = img
        import numpy as np
        frac = float(np.mean(np.all(img.view("u4") == ref.view("u4"), axis=-1)))
        rm = float(np.sqrt(np.mean((np.nan_to_num(img[..., :3]).astype("f8") - np.nan_to_num(ref[..., :3])) ** 2)))
        same = f"{frac:.6f} (rmse {rm:.2e})"
        print(f"flags={f} rep={rep} total={st['ms_device']:8.1f} ms  extend={st['ms_extend']:7.1f} connect={st['ms_connect']:7.1f} "
              f"shade={st['ms_shade']:7.1f} gmon={st['ms_gmon']:5.1f}  Msamples/s={w*h*spp/st['ms_device']*1e-3:7.1f} identical={same}", flush=True)
