import sys
sys.path.insert(0, '/root/repo')
from yart_amd import api, scenes
scene, p = scenes.sponza_class(480, 270, 64, 8, tex=256, sky=256)
ds = api.DeviceScene(scene, device=0, instrumented=True)
prev = None
n = 480*270*64
for d in range(1, 9):
    q = dict(p, depth=d)
    img, st = ds.render(q)
    cur = (st['lean_traversals'], st['shaded_hits'], st['rays'], st['traversals'])
    if prev: print(d, "extend rays at this bounce: %.3f of paths, shaded hits %.3f, all rays %.3f" % ((cur[0]-prev[0])/n, (cur[1]-prev[1])/n, (cur[2]-prev[2])/n))
    else: print(d, "extend %.3f shaded %.3f rays %.3f" % (cur[0]/n, cur[1]/n, cur[2]/n))
    prev = cur
