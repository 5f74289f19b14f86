import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
REF = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
tmp = tempfile.mkdtemp()
for seed in [int(x) for x in sys.argv[1:]]:
    spp, depth = 4 if seed % 3 else 16, 6 if seed % 4 else 12
    s, p = scenes.random_scene(seed, 64, 48, spp, depth)
    print(f"== seed {seed}: meshes {len(s.meshes)} nodes {len(s.nodes)} mats {len(s.materials)} tex {len(s.textures)} lights {len(s.lights)} tris {sum(len(m.faces) for m in s.meshes)} spp {spp} depth {depth} fnumber {p['fnumber']}")
    ds = api.DeviceScene(s, device=0, instrumented=True)
    for d in range(1, depth + 1):
        q = dict(p, depth=d)
        a, sa = ds.render(q, flags=0)
        b, sb = ds.render(q, flags=4)
        nd = int((a.view(np.uint32) != b.view(np.uint32)).any(-1).sum())
        keys = [k for k in sa if isinstance(sa[k], int) and sa[k] != sb[k] and not k.startswith("ms")]
        print(f" depth {d}: pixels differing {nd}; rays {sa['rays']} vs {sb['rays']}; counters that differ: " + ", ".join(f"{k} {sa[k]}/{sb[k]}" for k in keys[:12]))
        if nd:
            ys, xs = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(-1))
            for y, x in list(zip(ys, xs))[:6]:
                print(f"   pixel ({x},{y}) lean {a[y, x, :3]} general {b[y, x, :3]}")
            break
    ds.close()
