"""GPU-box measurement under `rocprofv3 --kernel-trace`: one render of the bench workload, whole (WORLD=1) or rank 0's share of
WORLD ranks (16x16 blocks), after a warm-up render — tools/shard_trace.sh compares the per-launch durations of the two."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
world = int(os.environ.get("WORLD", "1"))
scene, p = scenes.sponza_class(1920, 1080, 256, 8, tex=1024, sky=2048)
ds = api.DeviceScene(scene, device=0)
q = dict(p, shard_tile=16 if world > 1 else 0, max_batch_paths=1920 * 1080 * 256)    # (the whole frame: one batch)
for _ in range(2):
    _, st = ds.render(q, rank=0, world_size=world)
print(world, round(st["ms_device"], 2))
