"""examples/render_gltf.cpp — the reference's main.cpp flow (load GLB, add the .hdr environment, set the camera,
render, tonemap, write out.ppm) on the C++ mirror of the ABI."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT
from tests.paramfile import load_params

G = os.path.join(GOLDEN, "gltf")


@pytest.fixture(scope="module")
def example(built, tmp_path_factory):
    exe = os.path.join(tmp_path_factory.mktemp("ex"), "render_gltf")
    lib_dir = os.path.join(ROOT, "yart_amd")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "render_gltf.cpp"), "-L" + lib_dir, "-lyart_hip", "-Wl,-rpath," + lib_dir,
                    "-lpthread", "-o", exe], check=True)
    return exe


def _args(p, look):
    w, h = p["size"]
    return [str(v) for v in (w, h, p["spp"], p["depth"], *p["eye"], *p["target"], p["focal"], p["fnumber"], p["exposure"], look)]


def test_example_builds_and_fails_loudly_without_a_device(example, tmp_path):
    from yart_amd import api
    if api.lib().yart_hip_device_count() > 0:
        pytest.skip("a HIP device is visible")
    p = load_params(os.path.join(G, "gallery.txt"))
    r = subprocess.run([example, os.path.join(G, "gallery.glb"), os.path.join(G, "env_rle.hdr"), os.path.join(tmp_path, "o.ppm"),
                        *_args(p, -1)], capture_output=True, text=True)
    assert r.returncode == 3 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("look", [-1, 1])
def test_example_output_equals_the_python_path(example, tmp_path, look):
    from yart_amd import api
    p = load_params(os.path.join(G, "gallery.txt"))
    out = os.path.join(tmp_path, "o.ppm")
    subprocess.run([example, os.path.join(G, "gallery.glb"), os.path.join(G, "env_rle.hdr"), out, *_args(p, look)], check=True)
    raw = open(out, "rb").read()
    w, h = p["size"]
    header = b"P6\n%d %d\n255\n" % (w, h)
    assert raw.startswith(header)
    got = np.frombuffer(raw, np.uint8, offset=len(header)).reshape(h, w, 3)
    scene = api.DeviceScene(os.path.join(G, "gallery.glb"), device=0, env_hdr=os.path.join(G, "env_rle.hdr"))
    img, _ = scene.render(p)
    scene.close()
    name = {-1: None, 0: "none", 1: "golden", 2: "punchy"}[look]
    ldr, rgb8 = api.tonemap(img, name)
    if look >= 0:                                    # the example encodes the tonemapped frame without a second tonemap
        _, rgb8 = api.tonemap(ldr, None)
    assert np.array_equal(got, rgb8)
    if look < 0:                                     # and the linear frame is the reference's (golden)
        ref = np.fromfile(os.path.join(G, "gallery.f32"), np.float32).reshape(h, w, 4)
        _, want = api.tonemap(ref, None)
        assert np.array_equal(got, want)
