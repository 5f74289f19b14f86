import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from yart_amd import api, scenes
scene, p = scenes.sponza_class(1920, 1080, 256, 8)
ds = api.DeviceScene(scene, device=0)
for depth in (8, 4, 8, 4):
    q = dict(p, depth=depth)
    best = 1e9
    for rep in range(3):
        img, st = ds.render(q)
        if rep: best = min(best, st["ms_device"])
    print(f"depth {depth}: {best:.2f} ms, paths at bounce {[int(x) for x in st['paths_at_bounce'][:9]]}", flush=True)
for cap in (530841600,):
    for depth in (8, 4):
        q = dict(p, depth=depth, max_batch_paths=cap)
        best = 1e9
        for rep in range(3):
            img, st = ds.render(q)
            if rep: best = min(best, st["ms_device"])
        print(f"one batch, depth {depth}: {best:.2f} ms", flush=True)
