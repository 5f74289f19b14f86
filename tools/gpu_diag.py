"""GPU-box diagnostic: render the golden cases through the C ABI, dump framebuffers and
per-sample probe radiance next to the reference values (gpurun_out/diag_*.npz)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api
from tests import katlib
from tests.paramfile import load_params

out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
for case in sys.argv[1:] or ["cornell", "material", "cornell_waves"]:
    base = os.path.join(ROOT, "tests", "golden", case)
    p = load_params(base + ".txt")
    sc = api.DeviceScene(base + ".yscn", device=0)
    img, st = sc.render(p)
    ref = np.fromfile(base + ".f32", np.float32).reshape(img.shape)
    kat = katlib.load(base + ".kat.json")
    xys = [(x, y, s) for (x, y) in p["probe_pixels"] for s in range(p["spp"])]
    got, rays = sc.probe_samples(p, xys)
    rref = katlib.as_float(kat["radiance"]).reshape(-1, 3)
    d = np.abs(img[..., :3].astype(np.float64) - ref[..., :3])
    print(case, "rmse", np.sqrt((d ** 2).mean()), "max", d.max(), "pixels>1e-3", int((d.max(-1) > 1e-3).sum()),
          "identical", float(np.all(img.view(np.uint32) == ref.view(np.uint32), -1).mean()))
    ex = np.all(got.view(np.uint32) == rref.view(np.uint32), 1)
    cl = np.all(np.isclose(got, rref, rtol=1e-4, atol=1e-5, equal_nan=True), 1)
    print("  probe samples: exact", ex.mean(), "close", cl.mean(), "rays", rays, kat["probe_rays"][0])
    for i in np.nonzero(~cl)[0][:10]:
        print("   sample", xys[i], "gpu", got[i], "ref", rref[i])
    np.savez(os.path.join(out, f"diag_{case}.npz"), img=img, ref=ref, got=got, rref=rref)
    sc.close()
