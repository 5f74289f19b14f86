"""GPU-box A/B of library variants (tools/build_variant.sh NAME "-D...") on the bench scene: stage and kernel times, and every
variant's frame against the FIRST variant's (identical-pixel fraction, RMSE) — plus, with GOLDEN=1, against the committed
reference goldens. One subprocess per variant (the library handle is process-global).
   python tools/variant_ab.py base lazy trig ...      SIZE=1920x1080x64 TEX=1024 SKY=2048 REPS=2 FLAGS=0 SCENE=sponza|mclaren
A name may carry environment settings for its run: expsort:YART_EXP_SORT=3,YART_EXP_SORT_BITS=4"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(name):
    import numpy as np
    from yart_amd import api, scenes
    from tests.paramfile import load_params
    api.LIB_PATH = os.path.join(ROOT, "yart_amd", "_variants", name + ".so") if name != "PRODUCT" else api.LIB_PATH
    w, h, spp = (int(x) for x in os.environ.get("SIZE", "1920x1080x64").split("x"))
    tex, sky = int(os.environ.get("TEX", 1024)), int(os.environ.get("SKY", 2048))
    flags = int(os.environ.get("FLAGS", 0))
    if os.environ.get("SCENE", "sponza") == "mclaren":
        scene, p = scenes.mclaren_class(w, h, spp, 8, detail=1.0, tex=tex, sky=sky)
    else:
        scene, p = scenes.sponza_class(w, h, spp, 8, tex=tex, sky=sky)
    ds = api.DeviceScene(scene, device=0)
    best = None
    for rep in range(int(os.environ.get("REPS", 2)) + 1):
        img, st = ds.render(p, flags=flags)
        if rep and (best is None or st["ms_device"] < best["ms_device"]):
            best = st
    np.save(f"/tmp/variant_{name}.npy", img)
    out = {k: round(best[k], 2) for k in ("ms_device", "ms_extend", "ms_extend_lean", "ms_connect", "ms_shadow_lean", "ms_shade", "ms_shade_kernel")}
    ds.close()
    if os.environ.get("GOLDEN"):
        gold = {}
        for case in ("cornell", "material", "two_skies"):
            base = os.path.join(ROOT, "tests", "golden", case)
            g = api.DeviceScene(base + ".yscn", device=0)
            im, _ = g.render(load_params(base + ".txt"))
            ref = np.fromfile(base + ".f32", np.float32).reshape(im.shape)
            e = float(np.sqrt(np.mean((np.nan_to_num(im[..., :3]).astype("f8") - np.nan_to_num(ref[..., :3])) ** 2)))
            same = float(np.mean(np.all(im.view("u4") == ref.view("u4"), axis=-1)))
            gold[case] = {"rmse": e, "identical": round(same, 5)}
            g.close()
        out["golden"] = gold
    print("RESULT " + json.dumps(out), flush=True)


def main():
    import numpy as np
    names = sys.argv[1:]
    first = None
    for spec in names:
        n, _, envs = spec.partition(":")
        env = dict(os.environ, **dict(kv.split("=", 1) for kv in envs.split(",") if kv))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], capture_output=True, text=True, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if r.returncode != 0 or not line:
            print(f"{spec}: FAILED rc={r.returncode} {r.stderr.strip()[-400:]}", flush=True)
            continue
        res = json.loads(line[-1][7:])
        img = np.load(f"/tmp/variant_{n}.npy")
        if first is None:
            first = img
        same = float(np.mean(np.all(img.view("u4") == first.view("u4"), axis=-1)))
        e = float(np.sqrt(np.mean((np.nan_to_num(img[..., :3]).astype("f8") - np.nan_to_num(first[..., :3])) ** 2)))
        gold = res.pop("golden", None)
        print(f"{spec:14s} {json.dumps(res)}  vs {names[0]}: identical {same:.6f} rmse {e:.3e}" + (f"  goldens {json.dumps(gold)}" if gold else ""), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
