"""GPU-box experiment: every rank of 8 of the bench frame rendered alone, for several sizes of the sharding blocks (shard_tile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
scene, p = scenes.sponza_class(1920, 1080, 256, 8, tex=1024, sky=2048)
ds = api.DeviceScene(scene, device=0)
q = dict(p, max_batch_paths=1920 * 1080 * 256)
_, st = ds.render(q); _, st = ds.render(q)
full = st["ms_device"]
print(f"whole frame {full:.1f} ms -> /8 = {full / 8:.2f}", flush=True)
for tile in (8, 16, 32, 64, 128):
    worst = 0; tot = 0
    for rank in range(8):
        _, st = ds.render(dict(q, shard_tile=tile), rank=rank, world_size=8)
        _, st = ds.render(dict(q, shard_tile=tile), rank=rank, world_size=8)
        worst = max(worst, st["ms_device"]); tot += st["ms_device"]
    print(f"shard_tile {tile:3d}: slowest rank {worst:.2f} ms = {worst / (full / 8):.3f} x ideal, mean rank {tot / 8:.2f} ms", flush=True)
