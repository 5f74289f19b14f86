"""One render of the reduced C3 workload for counter collection under rocprofv3 --pmc.
Usage: [SCENE=mclaren] python3 tools/pmc_run.py FLAGS [W H SPP]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
if os.environ.get("YART_LIB") and os.environ["YART_LIB"] != "main":
    api.LIB_PATH = os.path.join(ROOT, "yart_amd", "_variants", os.environ["YART_LIB"] + ".so")
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
w, h, spp = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (480, 270, 32)
if os.environ.get("SCENE", "sponza") == "mclaren":
    scene, p = scenes.mclaren_class(w, h, spp, 8, detail=1.0, tex=int(os.environ.get("TEX", 256)), sky=int(os.environ.get("SKY", 256)))
else:
    scene, p = scenes.sponza_class(w, h, spp, 8, tex=int(os.environ.get("TEX", 256)), sky=int(os.environ.get("SKY", 256)))
ds = api.DeviceScene(scene, device=0)
img, st = ds.render(p, flags=flags)
print({k: st[k] for k in ("ms_device", "ms_extend", "ms_connect", "ms_shade", "traversals", "rays")})
