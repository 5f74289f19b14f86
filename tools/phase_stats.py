"""GPU-box experiment: phase statistics of the wave tracer (debug build libyart_hip_stats.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
api.LIB_COUNT_PATH = os.path.join(ROOT, "yart_amd", "libyart_hip_stats.so")
scene, p = scenes.sponza_class(960, 540, 64, 8, tex=256, sky=256)
ds = api.DeviceScene(scene, device=0, instrumented=True)
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 2
img, st = ds.render(p, flags=flags)
c = ds.debug_counters()
names = ["outer", "walk-node", "inner/pop", "tri", "slowpath", "leaf", "refill"] if flags == 2 else \
        ["inner/pop", "tri", "leaf-visit", "walk-step", "mask-box", "outer-round", "refill", "box-pair"]
print("ms extend %.1f connect %.1f" % (st["ms_extend"], st["ms_connect"]))
for k, n in enumerate(names):
    it, act = c[8 + 2 * k], c[9 + 2 * k]
    print(f"{n:10s} wave-iterations {it:12d}  active lanes {act:14d}  util {act / max(it, 1) / 64:.3f}")
