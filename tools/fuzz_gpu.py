"""GPU-box experiment: the comparison of tests/test_fuzz_scenes.py over a range of seeds — `fuzz_gpu.py FIRST LAST`.

Every seed: a random scene (yart_amd/scenes.py) rendered by the compiled reference (oracle/_ref/yart_ref) and by the library in
every pipeline variant of tests/test_gpu_parity.py::PIPELINE_FLAGS; frames compared word for word, ray counts as integers. One
line per difference, a summary at the end (profiles/r4_fuzz.txt holds the runs of round 4).

Families (environment):
  (none)          scenes.fuzz_case(seed) at 64x48: materials, textures, cut-outs, glass, soups, nested instances, lights, cameras;
                  some seeds with crowds of instance nodes, every ninth with coincident / degenerate geometry and extreme scales
  CROWD_ONLY=1    only the seeds with a crowd (70 .. 4300 nodes: the lean kernels' walks for 64 nodes and more)
  EXTRAS_ONLY=1   only the seeds with random_scene's `extras`
  SIZE=WxH        the same scenes at a larger frame (the reference then runs with its default workers)
  FRAMES=1        scenes.fuzz_frame_case(seed): random frame sizes, sample counts, wave schedules, tile sizes, batches, pools
  COMBOS=1        six random ORs of the pipeline flags per scene instead of each flag alone
"""
import faulthandler
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yart_amd import api, scenes                              # noqa: E402
from tests.test_gpu_parity import PIPELINE_FLAGS              # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
ENV = os.environ.get


def has_crowd(seed):          # (the rule of scenes.fuzz_case)
    return seed % 7 == 4 or seed % 13 == 7 or seed % 101 == 100 or seed % 211 == 210


def pipelines_of(seed):
    if not ENV("COMBOS"):
        return PIPELINE_FLAGS
    rng = np.random.RandomState(seed + 99991)
    bits = sorted(set(PIPELINE_FLAGS.values()) - {0, 1})
    return {f"flags {f}": f for f in (sum(b for b in bits if rng.rand() < 0.3) for _ in range(6))}


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    tmp = tempfile.mkdtemp()
    sp, pp, rf = f"{tmp}/s.yscn", f"{tmp}/p.txt", f"{tmp}/r.f32"
    bad = bad_rays = ref_hangs = frames = nan_frames = 0
    t0 = time.time()
    for seed in range(first, last):
        if (ENV("CROWD_ONLY") and not has_crowd(seed)) or (ENV("EXTRAS_ONLY") and seed % 9 != 5):
            continue
        if ENV("SIZE"):
            w, h = (int(v) for v in ENV("SIZE").split("x"))
            s, p = scenes.fuzz_case(seed, w, h)
        else:
            s, p = scenes.fuzz_frame_case(seed) if ENV("FRAMES") else scenes.fuzz_case(seed)
        s.save(sp)
        # one reference worker for the small frames: its workers can wait for a wave that has passed (tests/test_fuzz_scenes.py)
        scenes.write_params(pp, p, threads=None if ENV("SIZE") else 1)
        try:
            out = subprocess.run([REF, "render", sp, pp, rf], check=True, capture_output=True, text=True, timeout=120).stdout
        except subprocess.TimeoutExpired:
            print(f"seed {seed}: the REFERENCE did not finish within 120 s (skipped)", flush=True)
            ref_hangs += 1
            continue
        ref_rays = int(json.loads(out.strip().splitlines()[-1])["rays"])      # (the sum of its per-tile counts: ref_driver.cpp)
        ref = np.fromfile(rf, np.uint32)
        nan_frames += int(np.isnan(ref.view(np.float32)).any())
        ds = api.DeviceScene(s, device=0)
        for name, flags in pipelines_of(seed).items():
            faulthandler.dump_traceback_later(90, exit=True)                   # a render that does not return: say where, and stop
            img, st = ds.render(p, flags=flags)
            faulthandler.cancel_dump_traceback_later()
            frames += 1
            if int(st["rays"]) != ref_rays:
                bad_rays += 1
                print(f"RAYS seed {seed} / {name}: {st['rays']} rays, the reference counts {ref_rays}", flush=True)
            g = np.ascontiguousarray(img, np.float32).view(np.uint32).ravel()
            if not np.array_equal(ref, g):
                bad += 1
                print(f"MISMATCH seed {seed} / {name}: {(ref != g).sum()} of {ref.size} words differ", flush=True)
        ds.close()
        if seed % (1 if ENV("SIZE") else 25) == 0:
            print(f"seed {seed} done, {time.time() - t0:.0f} s", flush=True)
    print(f"seeds {first}..{last - 1}: {frames} frames, {bad} differ from the reference's; {bad_rays} ray counts differ; "
          f"{nan_frames} reference frames hold a NaN; the reference itself hung on {ref_hangs} seeds")


if __name__ == "__main__":
    main()
