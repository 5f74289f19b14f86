"""The device headers and the host side of the library (scene build, .yscn loader, sampler tables) compiled for
the host with AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool: the CPU build is
where they run). The binary is tests/hostsim with -fsanitize=address,undefined; it must stay clean on the self test,
the BVH build check and a golden render — and the render must still be the reference's frame bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT

SAN = os.path.join(ROOT, "tests", "hostsim", "_build", "hostsim_san")
SRC = [os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp"), os.path.join(ROOT, "yart_amd", "csrc", "_gen", "lut_data.cpp")]


@pytest.fixture(scope="module")
def san(built):
    newest = max(os.path.getmtime(os.path.join(dp, f)) for d in ("yart_amd/csrc", "tests/hostsim", "oracle")
                 for dp, _, fs in os.walk(os.path.join(ROOT, d)) for f in fs if f.endswith((".hpp", ".cpp", ".inc")))
    if not os.path.exists(SAN) or os.path.getmtime(SAN) < newest:
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-ffp-contract=off", "-o", SAN] + SRC + ["-lpthread"], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("no sanitizer runtime for g++ here: " + r.stderr[-200:])
    return SAN


def test_host_build_is_clean_under_asan_and_ubsan(san, tmp_path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    for args in (["selftest"], ["bvhcheck", os.path.join(GOLDEN, "material.yscn"), "3"]):
        r = subprocess.run([san] + args, capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, (args, r.stderr[-800:])
    out = os.path.join(tmp_path, "m.f32")
    base = os.path.join(GOLDEN, "material")
    r = subprocess.run([san, "render", base + ".yscn", base + ".txt", out], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-800:]
    assert np.array_equal(np.fromfile(out, np.uint32), np.fromfile(base + ".f32", np.uint32))


IMPORT_SAN = os.path.join(ROOT, "tests", "hostsim", "_build", "import_san")


@pytest.fixture(scope="module")
def import_san(built):
    src = os.path.join(ROOT, "tests", "hostsim", "import_san.cpp")
    newest = max(os.path.getmtime(p) for p in [src] + [os.path.join(ROOT, "yart_amd", "csrc", f) for f in
                                                       ("gltf_reader.hpp", "image_decode.hpp", "json_mini.hpp", "scene_file.hpp")])
    if not os.path.exists(IMPORT_SAN) or os.path.getmtime(IMPORT_SAN) < newest:
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-ffp-contract=off", "-o", IMPORT_SAN, src, "-lz", "-lpthread"], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("no sanitizer runtime for g++ here: " + r.stderr[-200:])
    return IMPORT_SAN


def _corrupt(data, rng):
    b = bytearray(data)
    mode = int(rng.integers(5))
    if mode == 0:
        b = b[:int(rng.integers(1, len(b)))]
    elif mode == 1:
        for _ in range(int(rng.integers(1, 10))):
            b[int(rng.integers(len(b)))] = int(rng.integers(256))
    elif mode == 2:
        k = int(rng.integers(max(1, len(b) - 4)))
        b[k:k + 4] = int(rng.choice([0, 0xffffffff, 0x7fffffff, 0x80000000])).to_bytes(4, "little")
    elif mode == 3:
        k, m = int(rng.integers(len(b))), int(rng.integers(1, 64))
        b[k:k + m] = bytes(rng.integers(0, 256, m).astype(np.uint8))
    else:
        k, m = int(rng.integers(len(b))), int(rng.integers(1, 200))
        del b[k:k + m]
    return bytes(b)


def _clean(r):
    return r.returncode in (0, 1) and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_importer_survives_corrupted_assets(import_san, tmp_path):
    """Untrusted input: truncated / bit-flipped / spliced GLB containers, PNG and JPEG streams inside valid containers, Radiance
    files — the importer either imports or refuses (an exception -> exit 1), under AddressSanitizer and UBSan, and never hangs.
    (5000 such files run once: 0 reports.)"""
    from tests import gltf_assets as ga
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096")
    gallery = open(os.path.join(GOLDEN, "gltf", "gallery.glb"), "rb").read()
    p3 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    n3 = np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    glb, hdr = os.path.join(tmp_path, "c.glb"), os.path.join(tmp_path, "e.hdr")
    for it in range(60):
        rng = np.random.default_rng(7000 + it)
        kind = it % 4
        args = [import_san, glb]
        if kind == 0:
            data = _corrupt(gallery, rng)
        elif kind == 3:
            h, w = int(rng.integers(1, 30)), int(rng.integers(1, 40))
            with open(hdr, "wb") as f:
                f.write(_corrupt(ga.hdr_encode(rng.integers(0, 256, (h, w, 4)).astype(np.uint8), rle=w >= 8), rng))
            data, args = gallery, args + [hdr]
        else:
            img, ext, _ = ga.random_image(int(rng.integers(1 << 30)))
            b = ga.GltfBuilder()
            tex = b.texture(b.image(_corrupt(img, rng), **({"mime": "image/jpeg"} if ext == "jpg" else {})))
            mat = b.material(pbrMetallicRoughness={"baseColorTexture": {"index": tex}})
            b.node(b.mesh([{"attributes": {"POSITION": b.accessor(p3, "VEC3"), "NORMAL": b.accessor(n3, "VEC3"),
                                           "TEXCOORD_0": b.accessor(p3[:, :2].copy(), "VEC2")}, "material": mat}]), root=True)
            b.write_glb(glb)
            data = open(glb, "rb").read()
        with open(glb, "wb") as f:
            f.write(data)
        r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=120)
        assert _clean(r), (it, kind, r.returncode, r.stderr[-800:])


def test_scene_loader_survives_corrupted_containers(san, tmp_path):
    """Corrupted .yscn containers (truncated, header counts / words / bytes overwritten) through the loader, the scene build and a
    one-sample render of the sanitizer build: loaded or refused, no report (1650 such files run once: 0 reports)."""
    from yart_amd import scenes
    s, p = scenes.fuzz_case(3, 16, 12)
    good = s.tobytes()
    pp, path = os.path.join(tmp_path, "p.txt"), os.path.join(tmp_path, "bad.yscn")
    scenes.write_params(pp, dict(p, spp=1, depth=2), threads=1)
    for it in range(40):
        rng = np.random.default_rng(9000 + it)
        b = bytearray(_corrupt(good, rng))
        if it % 5 == 0:                                      # a count of the header
            k = 8 + 4 * int(rng.integers(8))
            b[k:k + 4] = int(rng.choice([0, 1, 2 ** 31 - 1, 2 ** 32 - 1, int(rng.integers(0, 1000))])).to_bytes(4, "little")
        with open(path, "wb") as f:
            f.write(bytes(b))
        r = subprocess.run([san, "render", path, pp, os.path.join(tmp_path, "o.f32")], capture_output=True, text=True, timeout=120)
        assert r.returncode >= 0 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (it, r.returncode, r.stderr[-800:])


TSAN = os.path.join(ROOT, "tests", "hostsim", "_build", "hostsim_tsan")


@pytest.fixture(scope="module")
def tsan(built):
    newest = max(os.path.getmtime(os.path.join(dp, f)) for d in ("yart_amd/csrc", "tests/hostsim")
                 for dp, _, fs in os.walk(os.path.join(ROOT, d)) for f in fs if f.endswith((".hpp", ".cpp", ".inc")))
    if not os.path.exists(TSAN) or os.path.getmtime(TSAN) < newest:
        r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-o", TSAN] + SRC + ["-lpthread"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("no ThreadSanitizer runtime for g++ here: " + r.stderr[-200:])
    return TSAN


def test_threaded_host_code_is_clean_under_tsan(tsan, tmp_path):
    """SURVEY §5 "Race detection": the reference has none and carries races of exactly the class this build met twice (workers
    reading the wave number unlocked, tile-renderer.hpp:161-162; totalRays added up before the buffer lock, :217-218). The host
    code here that runs on several threads — the task-parallel SAH build (multi-threaded binning above 4096 triangles, subtree
    tasks below), concurrent callers of the scene loader / scene build, and the tile-threaded host path tracer
    of the test harness — under ThreadSanitizer: no report, same bytes as single-threaded."""
    from yart_amd import scenes
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    s, p = scenes.sponza_class(64, 36, 1, 4, detail=0.3, tex=32, sky=32)
    assert s.n_triangles > 3 * 4096                       # (the multi-threaded binning path is taken)
    sp, pp = str(tmp_path / "s.yscn"), str(tmp_path / "p.txt")
    s.save(sp); scenes.write_params(pp, p, threads=4)

    def clean(r):
        return r.returncode == 0 and "ThreadSanitizer" not in r.stderr

    r = subprocess.run([tsan, "bvhcheck", sp, "4"], capture_output=True, text=True, env=env, timeout=600)
    assert clean(r) and '"bvhcheck": "ok"' in r.stdout, r.stderr[-1500:]
    r = subprocess.run([tsan, "loadstress", sp, "4"], capture_output=True, text=True, env=env, timeout=600)
    assert clean(r) and '"loadstress": "ok"' in r.stdout, r.stderr[-1500:]
    # the tile-threaded host render (4 workers pulling tiles, as the reference's workers do) must give the golden frame
    base = os.path.join(GOLDEN, "cornell")
    gp = str(tmp_path / "g.txt")
    with open(base + ".txt") as f:
        lines = [ln for ln in f.read().splitlines() if not ln.startswith("threads")]
    with open(gp, "w") as f:
        f.write("\n".join(lines + ["threads 4"]) + "\n")
    out = str(tmp_path / "g.f32")
    r = subprocess.run([tsan, "render", base + ".yscn", gp, out], capture_output=True, text=True, env=env, timeout=900)
    assert clean(r), r.stderr[-1500:]
    assert np.array_equal(np.fromfile(out, np.uint32), np.fromfile(base + ".f32", np.uint32))
