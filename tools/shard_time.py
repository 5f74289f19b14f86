"""GPU-box experiment: per-rank render time of the bench workload under tile sharding (one GPU,
rank r of world w rendered alone) vs the ideal 1/w of the full-frame time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
scene, p = scenes.sponza_class(1920, 1080, 256, 8)
p = dict(p, shard_tile=int(os.environ.get("SHARD", 0)))
ds = api.DeviceScene(scene, device=0)
ds.render(p, rank=0, world_size=8)
for world in (4, 8):
    ts = []
    for rank in range(world if world > 1 else 1):
        t0 = time.perf_counter(); _, st = ds.render(p, rank=rank, world_size=world); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"world {world}: per-rank wall ms min {min(ts):.1f} max {max(ts):.1f} (ideal {ts and 0:.0f})", [round(t, 1) for t in ts], flush=True)
