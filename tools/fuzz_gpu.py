"""GPU-box experiment: tests/test_fuzz_scenes.py's comparison over a wider range of seeds (FIRST LAST), every pipeline variant of
tests/test_gpu_parity.py::PIPELINE_FLAGS, against the compiled reference (oracle/_ref/yart_ref). Prints one line per mismatch. CROWD_ONLY=1: only the seeds with 64 nodes and more. EXTRAS_ONLY=1: only the seeds with
random_scene's extras. COMBOS=1: random combinations of the pipeline flags. FRAMES=1: the second family
(scenes.fuzz_frame_case: random frame sizes, sample counts, wave schedules, tile sizes)."""
import faulthandler, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes
from tests.test_gpu_parity import PIPELINE_FLAGS
REF = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
first, last = int(sys.argv[1]), int(sys.argv[2])
tmp = tempfile.mkdtemp()
bad = 0; bad_rays = 0; ref_hangs = 0; frames = 0; nan_frames = 0; t0 = time.time()
for seed in range(first, last):
    if os.environ.get("CROWD_ONLY") and not (seed % 7 == 4 or seed % 13 == 7 or seed % 101 == 100 or seed % 211 == 210):
        continue                                   # (only the seeds scenes.fuzz_case gives a crowd of instance nodes)
    if os.environ.get("EXTRAS_ONLY") and seed % 9 != 5:
        continue                                   # (only the seeds with coincident duplicates, degenerate triangles, extreme scales)
    if os.environ.get("SIZE"):                      # SIZE=WxH: the first family at a larger frame (many tiles: the reference's default workers)
        w, h = (int(v) for v in os.environ["SIZE"].split("x"))
        s, p = scenes.fuzz_case(seed, w, h)
    else:
        s, p = scenes.fuzz_frame_case(seed) if os.environ.get("FRAMES") else scenes.fuzz_case(seed)
    sp, pp, rf = f"{tmp}/s.yscn", f"{tmp}/p.txt", f"{tmp}/r.f32"
    s.save(sp); scenes.write_params(pp, p, threads=None if os.environ.get("SIZE") else 1)     # (one worker: see _reference_frame's note in tests/test_fuzz_scenes.py)
    try:
        out = subprocess.run([REF, "render", sp, pp, rf], check=True, capture_output=True, text=True, timeout=120).stdout
        ref_rays = int(json.loads(out.strip().splitlines()[-1])["rays"])
    except subprocess.TimeoutExpired:
        print(f"seed {seed}: the REFERENCE did not finish within 120 s (skipped)", flush=True); ref_hangs += 1
        continue
    ref = np.fromfile(rf, np.uint32)
    nan_frames += int(np.isnan(ref.view(np.float32)).any())
    ds = api.DeviceScene(s, device=0)
    pipelines = PIPELINE_FLAGS
    if os.environ.get("COMBOS"):                   # COMBOS=1: six random ORs of the pipeline flags instead of each flag alone
        frng = np.random.RandomState(seed + 99991)
        bits = sorted(set(PIPELINE_FLAGS.values()) - {0, 1})
        pipelines = {}
        for _ in range(6):
            f = 0
            for b in bits:
                if frng.rand() < 0.3: f |= b
            pipelines[f"flags {f}"] = f
    for name, flags in pipelines.items():
        faulthandler.dump_traceback_later(90, exit=True)       # a render that does not return: say where, and stop
        print(f"seed {seed} {name}", file=open(os.path.join(tmp, "last"), "w"))
        img, st = ds.render(p, flags=flags)
        faulthandler.cancel_dump_traceback_later()
        if int(st["rays"]) != ref_rays:             # the reference's ray count (RenderData::totalRays): an integer result
            bad_rays += 1
            print(f"RAYS seed {seed} / {name}: {st['rays']} rays, the reference counts {ref_rays}", flush=True)
        g = np.ascontiguousarray(img, np.float32).view(np.uint32).ravel()
        frames += 1
        if not np.array_equal(ref, g):
            bad += 1
            print(f"MISMATCH seed {seed} / {name}: {(ref != g).sum()} of {ref.size} words differ", flush=True)
    ds.close()
    if seed % (1 if os.environ.get('SIZE') else 25) == 0: print(f"seed {seed} done, {time.time() - t0:.0f} s", flush=True)
print(f"seeds {first}..{last - 1}: {frames} frames of {len(PIPELINE_FLAGS)} pipelines, {bad} differ from the reference's; "
      f"{bad_rays} ray counts differ; {nan_frames} reference frames hold a NaN; the reference itself hung on {ref_hangs} seeds")
