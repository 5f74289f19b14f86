"""GPU-box measurement: cost of scenes with many scene-graph nodes (instanced meshes) — the lean kernels' per-ray pass over
the node boxes is linear in the node count (trace_lean.hpp part (A); chunked form from 64 nodes on)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
if os.environ.get("YART_LIB"): api.LIB_PATH = os.path.join(ROOT, "yart_amd", "_variants", os.environ["YART_LIB"] + ".so")
compact = os.environ.get("COMPACT") == "1"      # groups of instances that sit together (a group's box is small) instead of all over the room
for n in (16, 60, 70, 250, 1000, 4000):
    kw = dict(child_extent=0.35, group_extent=3.0, scale=(0.04, 0.12)) if compact else {}
    scene, p = scenes.instances(960, 540, 16, 4, n_instances=n, groups=max(2, n // 16), **kw)
    ds = api.DeviceScene(scene, device=0)
    fl = int(os.environ.get("FLAGS", 0))       # 65536: per-lane walk, 131072: top-level hierarchy, from 64 nodes on
    ds.render(p, flags=fl)
    img, st = ds.render(p, flags=fl)
    rays = st["rays"] if "rays" in st else 0
    print(f"{len(scene.nodes):5d} nodes: total {st['ms_device']:8.1f} ms  extend {st['ms_extend']:7.1f}  connect {st['ms_connect']:7.1f}  shade {st['ms_shade']:7.1f}  rays {rays}", flush=True)
