"""GPU-box measurement: where a wave of k_wf_shade spends its cycles (library variant built with -DYART_SHADE_REGIONS=1:
tools/build_variant.sh regions "-DYART_SHADE_REGIONS=1"). Usage: python tools/shade_regions.py [SIZE=WxHxSPP]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
api.LIB_PATH = os.path.join(ROOT, "yart_amd", "_variants", os.environ.get("YART_LIB", "regions") + ".so")
w, h, spp = (int(x) for x in os.environ.get("SIZE", "1920x1080x64").split("x"))
if os.environ.get("SCENE", "sponza") == "mclaren":
    scene, p = scenes.mclaren_class(w, h, spp, 8, detail=1.0, tex=1024, sky=2048)
else:
    scene, p = scenes.sponza_class(w, h, spp, 8, tex=1024, sky=2048)
ds = api.DeviceScene(scene, device=0)
L = ds._L
L.yart_hip_debug_shade_regions.argtypes = [C.c_void_p]
out = (C.c_uint64 * 48)()
img, st = ds.render(p, flags=0)                      # warm-up
assert L.yart_hip_debug_shade_regions(out) == 0
img, st = ds.render(p, flags=0)
assert L.yart_hip_debug_shade_regions(out) == 0
names = ["path state load", "miss: environment + MIS", "finalizeHit", "sampler: 2D + 1D + 1D", "frame + material / textures",
         "BSDF sample", "emission, throughput, new ray store", "sampler: NEE 1D + 2D", "light choice + environment sample",
         "BSDF f (light direction)", "BSDF pdf + shadow set-up store", "roulette, stores, queue appends",
         "tile grab + bucketing", "-", "-", "-"]
tot = sum(out[k] for k in range(16))
print(f"shade kernel {st['ms_shade_kernel']:.1f} ms (instrumented build), stage {st['ms_shade']:.1f} ms; wave cycles by region:")
for k in range(16):
    if out[16 + k]:
        print(f"{k:2d} {names[k]:38s} {100.0 * out[k] / tot:5.1f} %   visits {out[16 + k]:11d}   lanes/visit {out[32 + k] / out[16 + k]:5.1f}   cycles/visit {out[k] / out[16 + k]:8.0f}")
