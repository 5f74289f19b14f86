"""integration/hip-renderer.hpp — the yart::Renderer a maintainer adds to the reference tree — compiled against the
reference's own headers (oracle/_ref/yart_ref_hip, `make -C oracle ref_hip`; the binary travels to the GPU box) and
driven like src/main.cpp drives TileRenderer: same knobs, wave callbacks, the reference's host AgX object."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT
from tests.paramfile import load_params

G = os.path.join(GOLDEN, "gltf")
EXE = os.path.join(ROOT, "oracle", "_ref", "yart_ref_hip")
needs_exe = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/yart_ref_hip not built (needs the reference sources)")


@pytest.fixture(scope="module")
def adapter_exe():
    """On a GPU run the adapter binary (built here against the reference's headers, pushed prebuilt) must be there: fail, not skip."""
    from tests.conftest import _checker
    return _checker(EXE, "the reference-side adapter binary")


def run(tmp_path, params, look="-"):
    out = os.path.join(tmp_path, "a.f32")
    r = subprocess.run([EXE, os.path.join(G, "gallery.glb"), os.path.join(G, "env_rle.hdr"), params, out, look],
                       capture_output=True, text=True)
    return r, out


@needs_exe
def test_adapter_fails_loudly_without_a_device(built, tmp_path):
    from yart_amd import api
    if api.lib().yart_hip_device_count() > 0:
        pytest.skip("a HIP device is visible")
    r, _ = run(tmp_path, os.path.join(G, "gallery.txt"))
    assert r.returncode == 3 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
def test_adapter_render_equals_reference(adapter_exe, tmp_path):
    p = load_params(os.path.join(G, "gallery.txt"))
    w, h = p["size"]
    r, out = run(tmp_path, os.path.join(G, "gallery.txt"))
    assert r.returncode == 0, r.stderr
    assert "wave 0: 16 samples, 16 / 16 taken" in r.stdout
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    ref = np.fromfile(os.path.join(G, "gallery.f32"), np.float32).reshape(h, w, 4)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # with the reference's AgX object as `tonemapper`: its buffer holds what the reference's tonemap gives
    r, out = run(tmp_path, os.path.join(G, "gallery.txt"), "golden")
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    ref = np.fromfile(os.path.join(G, "gallery.agx_golden.f32"), np.float32).reshape(h, w, 4)
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all()


@pytest.mark.gpu
def test_adapter_waves_follow_the_reference_schedule(adapter_exe, tmp_path):
    """first_wave 2, max_wave 8 over 16 spp: waves of 2, 4, 8, 2 (tile-renderer.hpp:284-289), blended as one
    uninterrupted render of the library blends them."""
    from yart_amd import api, scenes
    p = dict(load_params(os.path.join(G, "gallery.txt")), first_wave=2, max_wave=8)
    params = os.path.join(tmp_path, "w.txt")
    scenes.write_params(params, p)
    r, out = run(tmp_path, params)
    assert r.returncode == 0, r.stderr
    waves = [l for l in r.stdout.splitlines() if l.startswith("wave")]
    assert [l.split(":")[1].split(",")[0].strip() for l in waves] == ["2 samples", "4 samples", "8 samples", "2 samples"]
    w, h = p["size"]
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    scene = api.DeviceScene(os.path.join(G, "gallery.glb"), device=0, env_hdr=os.path.join(G, "env_rle.hdr"))
    want, _ = scene.render(p)
    scene.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_adapter_tile_callbacks_and_multi_device(adapter_exe, tmp_path):
    """HipRenderer with onRenderTileComplete set (yart_hip_render_tiles: Renderer::TileData per finished tile, the
    renderer's buffer holding the tonemapped tile) and with a MultiDeviceScene (yart_hip_multi_render; GPU 0 named
    twice on a one-GPU box): both give the frame of the plain adapter render."""
    p = load_params(os.path.join(G, "gallery.txt"))
    w, h = p["size"]
    ref = np.fromfile(os.path.join(G, "gallery.f32"), np.float32).reshape(h, w, 4)
    out = os.path.join(tmp_path, "t.f32")
    base = [EXE, os.path.join(G, "gallery.glb"), os.path.join(G, "env_rle.hdr"), os.path.join(G, "gallery.txt"), out, "-"]
    r = subprocess.run(base + ["tiles=20000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tiles = [l for l in r.stdout.splitlines() if l.startswith("tile")]
    n_tiles = ((w + 63) // 64) * ((h + 63) // 64)
    assert len(tiles) == n_tiles and tiles[-1].startswith(f"tile {n_tiles}/{n_tiles}")
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    r = subprocess.run(base + ["tiles=262144", "devices=0,0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
