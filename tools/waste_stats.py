"""GPU-box experiment (count build, -DYART_COUNT_TRAVERSAL=1): box tests the lean kernels spend on rays they then hand to the
general kernels, against the box tests of the general kernels' own walks of those rays — the part of a retry walk a resume
(instead of a restart) could save. Usage: YART_LIB=cnt python tools/waste_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
if os.environ.get("YART_LIB"):
    api.LIB_PATH = os.path.join(ROOT, "yart_amd", "_variants", os.environ["YART_LIB"] + ".so")
w, h, spp = (int(x) for x in os.environ.get("SIZE", "1920x1080x16").split("x"))
scene, p = scenes.sponza_class(w, h, spp, 8, tex=1024, sky=2048)
ds = api.DeviceScene(scene, device=0, instrumented=not os.environ.get("YART_LIB"))
img, st = ds.render(p, flags=0)
c = ds.debug_counters()
retry_box = c[2] - c[6] - c[25]
print({"box_total": c[2], "lean_extend_box": c[6], "lean_shadow_box": c[25], "retry_kernels_box": retry_box,
       "retry_rays": (c[29], c[30]), "wasted_in_lean_extend": c[27], "wasted_in_lean_shadow": c[31],
       "wasted_over_retry_box": round((c[27] + c[31]) / max(1, retry_box), 4),
       "retry_box_per_ray": round(retry_box / max(1, c[29] + c[30]), 1),
       "wasted_box_per_ray": round((c[27] + c[31]) / max(1, c[29] + c[30]), 1)}, flush=True)
print({"resumed_extend": c[8], "of": c[29], "resumed_shadow": c[9], "of_shadow": c[30]}, flush=True)
