"""GPU-box A/B: camera rays (bounce 0) through the one-ray-per-lane lean kernel (debug bit 4096) against the
in-wave-replacement kernel (the default), at depth 1 (bounce 0 alone) and depth 8 (the whole frame)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
scene, p = scenes.sponza_class(1920, 1080, 64, 8, tex=1024, sky=2048)
ds = api.DeviceScene(scene, device=0)
for rep in range(2):
    for f in (0, 4096):
        for depth in (1, 8):
            img, st = ds.render(dict(p, depth=depth), flags=f)
            print(f"flags={f} depth={depth} rep={rep} extend={st['ms_extend']:7.1f} lean={st['ms_extend_lean']:7.1f} connect={st['ms_connect']:7.1f} total={st['ms_device']:8.1f}", flush=True)
