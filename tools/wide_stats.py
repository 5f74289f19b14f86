"""GPU-box experiment (counting build): the lean kernels' walk of their own 8-wide trees on the C3 (or SCENE=mclaren)
workload — node visits and triangle tests per ray next to the binary walk's box / triangle tests (flag 256), and the
rays handed to the general kernels by cause. Usage: python tools/wide_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes

w, h, spp = (int(x) for x in os.environ.get("SIZE", "960x540x16").split("x"))
TEX, SKY = int(os.environ.get("TEX", 256)), int(os.environ.get("SKY", 256))
if os.environ.get("SCENE", "sponza") == "mclaren":
    scene, p = scenes.mclaren_class(w, h, spp, 8, detail=float(os.environ.get("DETAIL", 1.0)), tex=TEX, sky=SKY)
else:
    scene, p = scenes.sponza_class(w, h, spp, 8, tex=TEX, sky=SKY)
ds = api.DeviceScene(scene, device=0, instrumented=True)
for f in (256, 0):
    img, st = ds.render(p, flags=f)
    le, ls = max(st["lean_traversals"], 1), max(st["shadow_lean_traversals"], 1)
    print(f"flags={f}: extend rays {le}  box/ray {st['lean_box_tests'] / le:.2f} tri/ray {st['lean_tri_tests'] / le:.2f} "
          f"wide nodes/ray {st['wide_extend_nodes'] / le:.2f} wide tris/ray {st['wide_extend_tris'] / le:.2f} "
          f"handed {[round(x / le, 5) for x in st['wide_extend_handed']]} retried {st['retry_extend_traversals'] / le:.4f}")
    print(f"          shadow rays {ls}  box/ray {st['shadow_lean_box_tests'] / ls:.2f} tri/ray {st['shadow_lean_tri_tests'] / ls:.2f} "
          f"wide nodes/ray {st['wide_shadow_nodes'] / ls:.2f} wide tris/ray {st['wide_shadow_tris'] / ls:.2f} "
          f"handed {[round(x / ls, 5) for x in st['wide_shadow_handed']]} retried {st['retry_shadow_traversals'] / ls:.4f}", flush=True)
