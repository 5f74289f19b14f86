"""GPU-box experiment: how many rays the lean kernels hand to the general ones, and what the retry passes cost, for a
library variant built with -DYART_COUNT_TRAVERSAL=1 (tools/build_variant.sh). Usage: YART_LIB=name python tools/retry_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
if os.environ.get("YART_LIB"):
    api.LIB_PATH = os.path.join(ROOT, "yart_amd", "_variants", os.environ["YART_LIB"] + ".so")
w, h, spp = (int(x) for x in os.environ.get("SIZE", "1920x1080x16").split("x"))
scene, p = scenes.sponza_class(w, h, spp, 8, tex=1024, sky=2048)
ds = api.DeviceScene(scene, device=0)
for rep in range(2):
    img, st = ds.render(p, flags=0)
print(os.environ.get("YART_LIB", "main"), {k: st[k] for k in ("traversals", "retry_extend_traversals", "retry_shadow_traversals",
      "ms_extend", "ms_extend_lean", "ms_connect", "ms_shadow_lean", "ms_shade", "ms_device")}, flush=True)
