"""SURVEY §8(f) rank 3 — the reference's BVH build (core/bvh.hpp:41-184, 273-347) on the device.

The device build (yart_amd/csrc/bvh_build_device.inc) must give the node array and the index permutation of the host
build byte for byte — and the host build is compared with the reference's own in tests/test_bvh_build.py and
tests/test_gpu_parity.py::test_bvh_matches_reference_on_device. The traversal order, hence every frame, depends on
this tree: an "equivalent" tree would not do."""
import numpy as np
import pytest

from yart_amd import api, scenes


def _soup(n, seed, extent=4.0, size=0.3):
    rng = np.random.RandomState(seed)
    c = (rng.rand(n, 1, 3) - 0.5) * extent
    p = (c + (rng.rand(n, 3, 3) - 0.5) * size).astype(np.float32).reshape(-1, 3)
    f = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return p, f


def _grid(nu, nv):
    u, v = np.meshgrid(np.linspace(0, 1, nu), np.linspace(0, 1, nv), indexing="ij")
    p = np.stack([u * 10 - 5, 0.3 * np.sin(7 * u) * np.cos(5 * v), v * 10 - 5], -1).astype(np.float32).reshape(-1, 3)
    i = (np.arange(nu - 1)[:, None] * nv + np.arange(nv - 1)[None, :]).reshape(-1)
    f = np.concatenate([np.stack([i, i + 1, i + nv], -1), np.stack([i + 1, i + nv + 1, i + nv], -1)]).astype(np.uint32)
    return p, f


def _cases():
    yield "one triangle", (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32))
    yield "soup 37", _soup(37, 1)
    yield "soup 5000", _soup(5000, 2)
    yield "soup 70000 (a root of more than 2048 triangles: the 1024-thread path)", _soup(70000, 3)
    yield "grid 160 x 120", _grid(160, 120)
    # degenerate input: identical triangles (one leaf of 200, a partition that moves everything to one side) ...
    p, f = _soup(1, 4)
    yield "200 copies of one triangle", (p, np.repeat(f, 200, axis=0))
    # ... clusters of identical centroids next to ordinary triangles, axis-aligned slabs, exact zeros in the bounds
    p1, f1 = _soup(300, 5)
    p2 = np.array([[0.001, 0, 0], [0.001, 1, 0], [0.001, 0, 1], [-0.001, 0, 0], [-0.001, 2, 0], [-0.001, 0, 2]], np.float32)
    f2 = np.array([[0, 1, 2]] * 40 + [[3, 4, 5]] * 40, np.uint32) + len(p1)
    yield "coincident clusters + zero bounds", (np.concatenate([p1, p2]), np.concatenate([f1, f2]))
    # faces with a fourth word (the scene format's material index): the stride is honoured
    p, f = _soup(900, 6)
    yield "stride 4", (p, np.concatenate([f, np.full((len(f), 1), 7, np.uint32)], 1))


@pytest.mark.gpu
@pytest.mark.parametrize("name,mesh", list(_cases()), ids=[n for n, _ in _cases()])
def test_device_build_is_byte_identical(name, mesh):
    pos, faces = mesh
    hn, hi, _ = api.bvh_build(pos, faces, device=None, threads=1)
    dn, di, _ = api.bvh_build(pos, faces, device=0)
    assert len(dn) == len(hn)
    assert np.array_equal(di, hi), "index permutation differs"
    assert np.array_equal(dn, hn), "node array differs"


@pytest.mark.gpu
def test_device_build_of_the_bench_scene_mesh():
    """The C3 scene's main mesh (264 k triangles) and the McLaren-class scene's (detail 0.25): byte-identical, and timed."""
    for scene in (scenes.sponza_class(64, 36, 1, 1, tex=64, sky=64)[0], scenes.mclaren_class(64, 36, 1, 1, detail=0.25, tex=64, sky=64)[0]):
        m = max(scene.meshes, key=lambda q: len(q.faces))
        hn, hi, hms = api.bvh_build(m.positions, m.faces, device=None, threads=0)
        dn, di, dms = api.bvh_build(m.positions, m.faces, device=0)
        print(f"{len(m.faces)} triangles: host {hms:.1f} ms, device {dms:.1f} ms, {len(dn)} nodes")
        assert np.array_equal(di, hi) and np.array_equal(dn, hn)


@pytest.mark.gpu
def test_device_build_refuses_nan():
    pos, faces = _soup(50, 7)
    pos[17, 1] = np.nan
    with pytest.raises(api.YartError):
        api.bvh_build(pos, faces, device=0)


@pytest.mark.gpu
def test_scene_with_device_built_bvhs_renders_the_same_frame():
    """Scenes get their BVHs from the device build by default; with YART_SCENE_HOST_BVH from the host builder. The node arrays
    the kernels traverse and the frame are the same, bit for bit (instanced scene: shared meshes, transformed nodes, alpha card)."""
    scene, p = scenes.material_test(96, 64, 8, 5)
    a = api.DeviceScene(scene, device=0, host_bvh=True)
    b = api.DeviceScene(scene, device=0)
    for m in range(len(scene.meshes)):
        na, ia = a.bvh(m)
        nb, ib = b.bvh(m)
        assert np.array_equal(na, nb) and np.array_equal(ia, ib)
    fa, _ = a.render(p)
    fb, _ = b.render(p)
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))


@pytest.mark.gpu
def test_device_build_fuzz():
    """80 random meshes — soups, duplicated triangles, triangles snapped to a coarse lattice (many equal centroids and
    bounds, exact zeros), slivers along one axis — of 1 to 6000 triangles: every one byte-identical to the host build."""
    rng = np.random.RandomState(20240607)
    for case in range(80):
        n = int(rng.choice([1, 2, 3, 7, 21, 22, 64, 65, 257, 300, 1000, 2049, 6000]))
        kind = case % 4
        p, f = _soup(n, 1000 + case, extent=float(rng.choice([0.5, 4.0, 50.0])), size=float(rng.choice([0.01, 0.3, 2.0])))
        if kind == 1:                                   # snapped to a lattice
            p = (np.round(p * 2.0) / 2.0).astype(np.float32)
        elif kind == 2:                                 # duplicates: each triangle repeated 1..5 times
            f = np.repeat(f, rng.randint(1, 6, size=len(f)), axis=0)[:max(n, 1)]
        elif kind == 3:                                 # slivers along x
            p[:, 1:] *= 1e-3
        hn, hi, _ = api.bvh_build(p, f, device=None, threads=1)
        dn, di, _ = api.bvh_build(p, f, device=0)
        assert np.array_equal(di, hi) and np.array_equal(dn, hn), f"case {case}: kind {kind}, {len(f)} triangles"
