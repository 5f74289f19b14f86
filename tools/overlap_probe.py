"""GPU-box experiment (round 5): do the launch tails of one batch overlap another batch's work when the two run on separate
streams? Two replicas of the bench scene on ONE device (api.MultiDeviceScene([0, 0]): a host thread and a stream per replica,
each renders half of the frame's pixel blocks) against the single handle that renders the same frame batch after batch.
Usage: python tools/overlap_probe.py [N replicas ...]   SIZE=1920x1080x256"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from yart_amd import api, scenes

w, h, spp = (int(x) for x in os.environ.get("SIZE", "1920x1080x256").split("x"))
scene, p = scenes.sponza_class(w, h, spp, 8)
ds = api.DeviceScene(scene, device=0)
for cap, tag in ((0, "default 2^28"), (w * h * spp, "one batch"), (w * h * spp // 4 + 64, "4 batches")):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); ref, st = ds.render(dict(p, max_batch_paths=cap)); dt = time.perf_counter() - t0
        if rep: best = min(best, st["ms_device"])
    print(f"single handle, {tag}: {best:.1f} ms device ({w * h * spp / best * 1e-3:.1f} Msamples/s), wall of the last call {dt * 1e3:.1f} ms", flush=True)
ds.close()
for n in [int(a) for a in sys.argv[1:]] or [2, 4]:
    m = api.MultiDeviceScene(scene, [0] * n)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); img, st = m.render(dict(p, shard_tile=16)); dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    same = bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32)))
    print(f"{n} replicas on one device, concurrent streams: wall {best * 1e3:.1f} ms incl. merge + copy-out ({w * h * spp / best * 1e-6:.1f} Msamples/s), "
          f"slowest replica's device time {st['ms_device']:.1f} ms, frame identical {same}", flush=True)
    m.close()
