"""GPU-box experiment: the C3 frame rendered in 1, 2, 4, 8, 16 batches (YartRenderParams.max_batch_paths)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
scene, p = scenes.sponza_class(1920, 1080, 256, 8, tex=1024, sky=2048)
ds = api.DeviceScene(scene, device=0)
total = 1920 * 1080 * 256
for rep in range(2):
    for nb in [int(x) for x in os.environ.get("BATCHES", "1,2,4,8,16,32").split(",")]:
        q = dict(p, max_batch_paths=0 if nb == 1 else total // nb + 256)
        img, st = ds.render(q)
        print(f"batches={nb:2d} rep={rep} total={st['ms_device']:8.1f} ms extend={st['ms_extend']:7.1f} connect={st['ms_connect']:7.1f} shade={st['ms_shade']:7.1f}", flush=True)
