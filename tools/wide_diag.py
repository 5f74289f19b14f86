"""GPU-box diagnostic: frames of the 4-wide lean walk (YART_FLAG_WIDE_BVH) against the binary walk, scene by scene."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
from tests.paramfile import load_params

WF = int(os.environ.get("WIDE_FLAGS", 128))
def cmp(tag, ds, p):
    a, sa = ds.render(p, flags=0)
    b, sb = ds.render(p, flags=WF)
    same = float(np.mean(np.all(a.view(np.uint32) == b.view(np.uint32), axis=-1)))
    e = float(np.sqrt(np.mean((np.nan_to_num(a[..., :3]).astype(np.float64) - np.nan_to_num(b[..., :3])) ** 2)))
    print(f"{tag}: identical {same:.5f} rmse {e:.3e} rays {sa['rays']} vs {sb['rays']}", flush=True)

G = os.path.join(ROOT, "tests", "golden")
for case in ("cornell", "material"):
    ds = api.DeviceScene(os.path.join(G, case + ".yscn"), device=0)
    cmp(case, ds, load_params(os.path.join(G, case + ".txt")))
    for depth in (1, 2):
        cmp(f"{case} depth {depth}", ds, dict(load_params(os.path.join(G, case + ".txt")), depth=depth))
    ds.close()
s, p = scenes.heightfield(256, 256, 8, 4)
ds = api.DeviceScene(s, device=0); cmp("heightfield", ds, p); cmp("heightfield depth 1", ds, dict(p, depth=1)); ds.close()
s, p = scenes.instances(96, 96, 4, 4, n_instances=20)
ds = api.DeviceScene(s, device=0); cmp("instances", ds, p); ds.close()
s, p = scenes.sponza_class(240, 136, 16, 8, tex=256, sky=256)
ds = api.DeviceScene(s, device=0)
for depth in (1, 2, 8):
    cmp(f"sponza depth {depth}", ds, dict(p, depth=depth))
ds.close()
