"""GPU-box experiment: stage times of one rank's share of the bench workload (rank 0 of `world`, 16x16 blocks as bench.py
deals them) against 1/world of the whole frame's — where the strong-scaling loss of DESIGN §6 sits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
scene, p = scenes.sponza_class(1920, 1080, 256, 8, tex=1024, sky=2048)
ds = api.DeviceScene(scene, device=0)
keys = ("ms_device", "ms_extend", "ms_extend_lean", "ms_connect", "ms_shadow_lean", "ms_shade", "ms_shade_kernel", "ms_gmon")
_, full = ds.render(p)
_, full = ds.render(p)
print("whole frame   ", {k: round(full[k], 1) for k in keys}, flush=True)
for world in (2, 4, 8):
    q = dict(p, shard_tile=16)
    t0 = time.perf_counter(); _, st = ds.render(q, rank=0, world_size=world); wall = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); _, st = ds.render(q, rank=0, world_size=world); wall = (time.perf_counter() - t0) * 1e3
    print(f"rank 0 of {world}: wall {wall:.1f} ms", {k: f"{st[k]:.1f} ({st[k] * world / full[k]:.2f}x)" for k in keys}, flush=True)
