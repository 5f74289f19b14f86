"""BVH leaves of more than 31 triangles (the traversal-stack link word has 5 bits of leaf span).
The reference keeps a node as a leaf of any span when its split is degenerate (core/bvh.hpp:159-161); the scene
stacks 31 / 32 / 33 / 40 / 64 coincident quads, so such leaves end up as near AND far children. CPU: the device
headers compiled for the host (hostsim) against the compiled reference / the oracle; GPU: every pipeline."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ORACLE_BIN, REF_BIN


def _scene(tmp_path, spp=4):
    from yart_amd import scenes
    s, p = scenes.stacked_leaves(96, 96, spp, 4)
    sp, pp = str(tmp_path / "s.yscn"), str(tmp_path / "p.txt")
    s.save(sp); scenes.write_params(pp, p)
    return s, p, sp, pp


def _checker():
    exe = REF_BIN if os.path.exists(REF_BIN) else ORACLE_BIN
    if not os.path.exists(exe):
        pytest.skip("neither oracle/_ref/yart_ref nor the oracle restatement is built")
    return exe


def test_big_leaves_on_host(hostsim, tmp_path):
    _, _, sp, pp = _scene(tmp_path)
    info = json.loads(subprocess.run([hostsim, "bvhcheck", sp, "4"], check=True, capture_output=True, text=True).stdout)
    assert info["max_leaf_span"] >= 64, info          # the scene does produce the leaves in question
    ref, got = str(tmp_path / "ref.f32"), str(tmp_path / "got.f32")
    subprocess.run([_checker(), "render", sp, pp, ref], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([hostsim, "render", sp, pp, got], check=True, stdout=subprocess.DEVNULL)
    a, b = np.fromfile(ref, np.uint32), np.fromfile(got, np.uint32)
    assert np.array_equal(a, b), f"{(a != b).sum()} words differ"


@pytest.mark.gpu
def test_big_leaves_on_device(built, tmp_path):
    from yart_amd import api
    from tests.test_gpu_parity import PIPELINE_FLAGS
    s, p, sp, pp = _scene(tmp_path, spp=8)
    ref = str(tmp_path / "ref.f32")
    subprocess.run([_checker(), "render", sp, pp, ref], check=True, stdout=subprocess.DEVNULL)
    scene = api.DeviceScene(s, device=0)
    want = None
    for name, flags in PIPELINE_FLAGS.items():
        img, _ = scene.render(p, flags=flags)
        if want is None:
            want = np.fromfile(ref, np.float32).reshape(img.shape)
        same = float(np.mean(np.all(img.view(np.uint32) == want.view(np.uint32), axis=-1)))
        print(f"stacked leaves / {name}: identical_pixels={same:.4f}")
        assert same > 0.999, name
    scene.close()
