"""BASELINE.json configs on the GPU against the CPU checker ("what's weak" 1-3): the bench scene itself
with every pipeline, the C4 sampler state (1024 spp, log2spp = 10) and the C5 scene at full detail on windows of
the full-size frames, several batches per wave with path-state compaction on, and the reference's per-ray hit
records against the device traversal.

Windows of a frame: the library renders only the pixel blocks of rank r of N (YartRenderParams.rank / world_size /
shard_tile); the oracle restatement takes the same three numbers (oracle/params.hpp shard_*), so a full-size camera
and sampler state is checked on a few thousand pixels scattered over the frame in seconds."""
import ctypes
import json
import os
import subprocess

import numpy as np
import pytest

from tests import katlib
from tests.conftest import GOLDEN, ORACLE_BIN, REF_BIN, bit_identical_or_drift
from tests.paramfile import load_params
from tests.test_gpu_parity import PIPELINE_FLAGS, RMSE_TOL, rmse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(built):
    from yart_amd import api
    assert api.lib().yart_hip_device_count() > 0, "no HIP device: the GPU tests need the real kernels"
    return api


def _check(tmp_path, s, p, exe, shard=None):
    """Render (s, p) with the CPU checker `exe`; returns (frame, info)."""
    from yart_amd import scenes
    sp, pp, out = tmp_path / "c.yscn", tmp_path / "c.txt", tmp_path / "c.f32"
    s.save(sp)
    q = dict(p)
    if shard:
        q.update(shard_rank=shard[0], shard_world=shard[1], shard_tile=shard[2])
    scenes.write_params(pp, q)
    r = subprocess.run([exe, "render", str(sp), str(pp), str(out)], check=True, capture_output=True, text=True)
    w, h = p["size"]
    return np.fromfile(out, np.float32).reshape(h, w, 4), json.loads(r.stdout.strip().splitlines()[-1])


def _compare(img, ref, tag):
    """Asserts bit-identity with the checker's frame (conftest.bit_identical_or_drift); returns (identical fraction, rmse)."""
    return bit_identical_or_drift(img, ref, tag)


@pytest.mark.usefixtures("any_checker")
def test_bench_scene_vs_reference_every_pipeline(api, tmp_path):
    """The scene bench.py times (264 k triangles, alpha cut-outs -> retry queues, thin glass, 15-node graph with
    nested transforms, env light) against the compiled reference — every pipeline variant, not one against another."""
    from yart_amd import scenes
    s, p = scenes.sponza_class(240, 136, 16, 8, tex=256, sky=256)
    exe = REF_BIN if os.path.exists(REF_BIN) else ORACLE_BIN
    ref, info = _check(tmp_path, s, p, exe)
    scene = api.DeviceScene(s, device=0)
    for name, flags in PIPELINE_FLAGS.items():
        img, st = scene.render(p, flags=flags)
        same, e = _compare(img, ref, f"sponza_class 240x136x16 / {name} vs {os.path.basename(exe)}")
        assert int(st["rays"]) == int(info["rays"]), name      # ray counts are integers and the frames are bit-identical: exact
    scene.close()


@pytest.mark.usefixtures("oracle_bin")
def test_c4_1024spp_window_of_the_full_frame(api, tmp_path):
    """BASELINE configs[3]: Sponza-class 1920x1080 at 1024 spp (log2spp = 10, 32-bit Morton sample index, GMoN with
    15 buckets of 68-69 samples) — 8x8 pixel blocks scattered over the whole frame (rank 11 of 2000)."""
    from yart_amd import scenes
    s, p = scenes.sponza_class(1920, 1080, 1024, 8, tex=256, sky=256)
    shard = (11, 2000, 8)
    ref, info = _check(tmp_path, s, p, ORACLE_BIN, shard)
    scene = api.DeviceScene(s, device=0)
    q = dict(p, shard_tile=shard[2])
    img, st = scene.render(q, rank=shard[0], world_size=shard[1])
    assert st["samples"] == info["pixels"] * 1024 and info["pixels"] >= 900
    assert np.array_equal(img[..., 3] == 1.0, ref[..., 3] == 1.0), "library and oracle dealt different pixel blocks"
    same, e = _compare(img, ref, f"sponza_class 1080p x 1024 spp, {info['pixels']} pixels")
    mega, _ = scene.render(q, rank=shard[0], world_size=shard[1], flags=PIPELINE_FLAGS["megakernel"])
    assert np.array_equal(mega.view(np.uint32), img.view(np.uint32))
    # the whole C4 frame on one GPU (2.12 G paths, several batches), and as the 8 ranks of the BASELINE configuration
    # dealt 16-pixel blocks: the window's pixels are the oracle's either way, the 8 shares are disjoint and complete
    full, st = scene.render(p)
    assert st["samples"] == 1920 * 1080 * 1024 and np.isfinite(full).all()
    mask = ref[..., 3] == 1.0
    _compare(full[mask], ref[mask], "the window's pixels in the full 1024-spp frame")
    print(f"sponza_class 1920x1080x1024 full frame: {1920 * 1080 * 1024 / st['ms_device'] * 1e-3:.1f} Msamples/s")
    acc = np.zeros_like(full)
    for r in range(8):
        part, _ = scene.render(dict(p, shard_tile=16), rank=r, world_size=8)
        acc += part
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))
    scene.close()


@pytest.mark.usefixtures("oracle_bin")
def test_c3_256spp_window_of_the_bench_frame(api, tmp_path):
    """The TIMED configuration itself (BASELINE configs[2] as bench.py builds it: 1920x1080, 256 spp — log2spp = 8, four
    sample digits, two of them hashed per draw —, 8 bounces, 1024^2 textures, 2048^2 sky) against the oracle on 8x8 blocks
    scattered over the frame (rank 5 of 2000): default pipeline, megakernel and the unbucketed shade queue."""
    from yart_amd import scenes
    s, p = scenes.sponza_class(1920, 1080, 256, 8, tex=1024, sky=2048)
    shard = (5, 2000, 8)
    ref, info = _check(tmp_path, s, p, ORACLE_BIN, shard)
    scene = api.DeviceScene(s, device=0)
    q = dict(p, shard_tile=shard[2])
    for name in ("wavefront", "megakernel", "wavefront+no_shade_sort"):
        img, st = scene.render(q, rank=shard[0], world_size=shard[1], flags=PIPELINE_FLAGS[name])
        assert st["samples"] == info["pixels"] * 256 and info["pixels"] >= 900
        assert np.array_equal(img[..., 3] == 1.0, ref[..., 3] == 1.0), "library and oracle dealt different pixel blocks"
        _compare(img, ref, f"bench scene 1080p x 256 spp, {info['pixels']} pixels / {name}")
    scene.close()


@pytest.mark.usefixtures("oracle_bin")
def test_c5_full_detail_window_of_the_4k_frame(api, tmp_path):
    """BASELINE configs[4]: McLaren-class at detail 1 (1.05 M triangles; clearcoat, thin + refractive dielectric,
    chrome; f/2.8) at 3840x2160, 512 spp, 8 bounces — 8x8 blocks scattered over the frame (rank 7 of 4000), default
    pipeline and the material-bucketed shade queue."""
    from yart_amd import scenes
    s, p = scenes.mclaren_class(3840, 2160, 512, 8, detail=1.0, tex=256, sky=256)
    assert s.n_triangles > 1000000
    shard = (7, 4000, 8)
    ref, info = _check(tmp_path, s, p, ORACLE_BIN, shard)
    scene = api.DeviceScene(s, device=0)
    q = dict(p, shard_tile=shard[2])
    for name in ("wavefront", "wavefront+no_shade_sort"):
        img, st = scene.render(q, rank=shard[0], world_size=shard[1], flags=PIPELINE_FLAGS[name])
        assert st["samples"] == info["pixels"] * 512
        same, e = _compare(img, ref, f"mclaren_class 4K x 512 spp, {info['pixels']} pixels / {name}")
    # the WHOLE job on one GPU (8.3 M pixels x 512 spp = 4.25 G paths: several batches of what the memory holds, 1 GiB paths
    # at most): every pixel finite with alpha 1, and the window's pixels are still the oracle's
    full, st = scene.render(p)
    assert st["samples"] == 3840 * 2160 * 512
    assert np.isfinite(full).all() and np.all(full[..., 3] == 1.0)
    mask = ref[..., 3] == 1.0
    same, _ = _compare(full[mask], ref[mask], "the window's pixels in the full 4K frame")
    print(f"mclaren_class 3840x2160x512 full frame: {3840 * 2160 * 512 / st['ms_device'] * 1e-3:.1f} Msamples/s, "
          f"{st['ms_device'] / 1e3:.2f} s, window pixels identical {same:.5f}")
    scene.close()


def test_many_batches_equal_one_batch(api):
    """YartRenderParams.max_batch_paths: a wave rendered in many batches (chunk < pixels of the rank; compaction and
    tail states on, sampler tables indexed through pixBase) is the one-batch frame bit for bit — bench scene, two
    progressive waves, also when the batch is smaller than one pixel's samples of the wave."""
    from yart_amd import scenes
    s, p = scenes.sponza_class(240, 136, 16, 8, tex=256, sky=256)
    p = dict(p, first_wave=4, max_wave=12)
    scene = api.DeviceScene(s, device=0)
    one, st1 = scene.render(p)
    for cap, flags in ((100000, 0), (37000, 0), (37000, PIPELINE_FLAGS["wavefront+no_compaction"]), (200000, 1)):
        many, st = scene.render(dict(p, max_batch_paths=cap), flags=flags)
        assert np.array_equal(one.view(np.uint32), many.view(np.uint32)), (cap, flags)
        assert st["rays"] == st1["rays"] and st["samples"] == st1["samples"]
    scene.close()
    s, p = scenes.cornell(20, 12, 16, 4)                       # one pixel per batch (the cap is below one pixel's samples)
    scene = api.DeviceScene(s, device=0)
    one, _ = scene.render(p)
    many, _ = scene.render(dict(p, max_batch_paths=5))
    assert np.array_equal(one.view(np.uint32), many.view(np.uint32))
    scene.close()


@pytest.mark.parametrize("case", ["cornell", "material", "cornell_waves"])
def test_hit_records_vs_reference_kat(api, case):
    """The reference's per-ray hit records (KAT sections hit_rays / hits_i / hits_f: the centre-of-pixel ray of every
    probe pixel through RayIntegrator::testNode, oracle/ref_driver.cpp: did-hit, triangle, light index, back side,
    then t, uv, p, n, tangent) against the device's closest-hit traversal + finalizeHit (yart_hip_probe_hits)."""
    base = os.path.join(GOLDEN, case)
    kat = katlib.load(base + ".kat.json")
    rays = katlib.as_float(kat["hit_rays"]).reshape(-1, 6)
    hi = np.asarray(kat["hits_i"], np.int64).reshape(len(rays), 4)
    hf = katlib.as_float(kat["hits_f"]).reshape(len(rays), 12)
    scene = api.DeviceScene(base + ".yscn", device=0)
    out = scene.probe_hits(rays)
    hit = out[:, 0] > 0.5
    assert np.array_equal(hit, hi[:, 0] != 0) and hit.any()
    assert np.array_equal(out[hit, 13].astype(np.int64), hi[hit, 1]), "triangle index"
    assert np.array_equal(out[hit, 14].astype(np.int64), hi[hit, 2]), "light index"
    assert np.array_equal(out[hit, 15].astype(np.int64), hi[hit, 3]), "back side"
    # t, p, n, tangent: the device evaluates the reference's expressions in its order -> the same bits
    got = np.ascontiguousarray(np.concatenate([out[hit, 1:2], out[hit, 4:13]], axis=1))
    want = np.ascontiguousarray(np.concatenate([hf[hit, 0:1], hf[hit, 3:12]], axis=1))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    scene.close()


def test_tile_callback_reports_every_block_once_per_wave(api):
    """yart_hip_render_tiles (Renderer::onRenderTileComplete, renderer.hpp:40-50, tile-renderer.hpp:243-262): with
    batches of at most ~two tiles, every tile of every wave is reported exactly once, in Morton order, with the
    frame holding that tile's blended pixels at the time of the call; the final frame is the plain render; stopping
    from a tile callback returns the frame with exactly the tiles reported so far blended."""
    from yart_amd import dist as yd
    base = os.path.join(GOLDEN, "cornell")           # 128 x 128: four 64-pixel tiles, 16 spp
    p = dict(load_params(base + ".txt"), first_wave=8, max_wave=8, max_batch_paths=2 * 64 * 64 * 8 + 7)   # waves of 8 + 8 spp
    scene = api.DeviceScene(base + ".yscn", device=0)
    plain, _ = scene.render(p)
    first_wave, _ = scene.render(dict(p, stop_sample=8))
    seen, frames = [], []

    tile_rays = []

    def on_tile(frame, t):
        seen.append((t["wave"], t["x"], t["y"], t["width"], t["height"], t["index"], t["total"], t["samples_taken"]))
        frames.append(frame[t["y"]:t["y"] + t["height"], t["x"]:t["x"] + t["width"]].copy())
        tile_rays.append((t["wave"], t["rays"]))
    waves, wave_rays = [], []
    img, st, aborted = scene.render_tiles(p, on_tile, lambda f, info: (waves.append(info["wave"]), wave_rays.append(info["rays"])) and None)
    assert not aborted and np.array_equal(img.view(np.uint32), plain.view(np.uint32))
    assert waves == [0, 1] and len(seen) == 8
    # Renderer::TileData.rays (renderer.hpp:40-50): every block reports its own rays of the wave; they sum to the wave's count
    for w in range(2):
        mine = [r for ww, r in tile_rays if ww == w]
        assert all(r > 0 for r in mine) and sum(mine) == wave_rays[w], (w, mine, wave_rays)
    assert sum(r for _, r in tile_rays) == st["rays"]
    order = [(0, 0), (64, 0), (0, 64), (64, 64)]     # Morton order of the tile coordinates
    for w in range(2):
        got = seen[4 * w:4 * w + 4]
        assert [(g[1], g[2]) for g in got] == order
        assert [g[5] for g in got] == [1, 2, 3, 4] and all(g[0] == w and g[6] == 4 and g[3] == g[4] == 64 for g in got)
        assert all(g[7] == (8, 16)[w] for g in got)
    for k, (x, y) in enumerate(order):               # wave 0's tiles hold the first wave's pixels, wave 1's the final ones
        assert np.array_equal(frames[k].view(np.uint32), first_wave[y:y + 64, x:x + 64].view(np.uint32))
        assert np.array_equal(frames[4 + k].view(np.uint32), plain[y:y + 64, x:x + 64].view(np.uint32))
    # stop at the third tile of the first wave: batches hold two tiles, so the second batch has finished -> 4 tiles in
    count = []
    img2, st2, aborted2 = scene.render_tiles(p, lambda f, t: count.append(1) or len(count) >= 3)
    assert aborted2 and np.array_equal(img2.view(np.uint32), first_wave.view(np.uint32))
    # sharded: a rank reports only its own blocks (16-pixel blocks, rank 1 of 3)
    mine = []
    q = dict(p, shard_tile=16, first_wave=16, max_wave=16)
    part, _, _ = scene.render_tiles(q, lambda f, t: mine.append((t["x"], t["y"])) and None, rank=1, world_size=3)
    mask = yd.pixel_mask(128, 128, 16, 1, 3)
    assert len(mine) == mask.sum() // 256 and all(mask[y, x] for x, y in mine)
    assert np.array_equal(part[..., 3] == 1.0, mask)
    scene.close()


def test_multi_device_entry_equals_single_device(api):
    """yart_hip_multi_*: replicas + one host thread each + slab merge. On a one-GPU box the device list names GPU 0
    several times (the transport is then a device-to-device copy, everything else — dealing of the blocks, threads,
    pack / scatter — is what an 8-GPU node runs, where ncclSend / ncclRecv carry the slabs): the merged frame is the
    single-device frame bit for bit, with whole tiles and with 16-pixel blocks, one wave and several, and when the call
    is itself one rank of two processes."""
    from yart_amd import scenes
    s, p = scenes.sponza_class(240, 136, 8, 6, tex=128, sky=128)
    p = dict(p, max_batch_paths=1 << 18)             # the replicas of the rehearsal share one GPU's memory
    single = api.DeviceScene(s, device=0)
    want, st1 = single.render(p)
    multi = api.MultiDeviceScene(s, [0, 0, 0])
    assert multi.n_devices == 3
    got, st = multi.render(p)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert st["samples"] == st1["samples"] and st["rays"] == st1["rays"]
    q = dict(p, shard_tile=16, first_wave=2, max_wave=4)
    want2, _ = single.render(q)
    got2, _ = multi.render(q)
    assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32))
    halves = [multi.render(q, rank=r, world_size=2)[0] for r in range(2)]
    assert np.array_equal((halves[0] + halves[1]).view(np.uint32), want2.view(np.uint32))
    # the progressive form: wave callbacks whatever the number of devices (tile-renderer.hpp:243-282), every block reported once
    # per wave with its own ray count; the frame after each wave is the single device's after that wave
    waves, tiles = [], []
    want_w0, _ = single.render(dict(q, stop_sample=2))
    got3, st3, ab = multi.render_tiles(q, lambda f, t: tiles.append((t["wave"], t["x"], t["y"], t["rays"], t["index"], t["total"])) and None,
                                       lambda f, info: waves.append((info["wave"], info["samples_taken"], info["rays"], f.copy())) and None)
    assert not ab and np.array_equal(got3.view(np.uint32), want2.view(np.uint32))
    assert [(w, t) for w, t, _, _ in waves] == [(0, 2), (1, 6), (2, 8)]
    assert np.array_equal(waves[0][3].view(np.uint32), want_w0.view(np.uint32))
    n_blocks = -(-240 // 16) * -(-136 // 16)
    for w in range(3):
        mine = [t for t in tiles if t[0] == w]
        assert len(mine) == n_blocks and len({(t[1], t[2]) for t in mine}) == n_blocks and [t[4] for t in mine] == list(range(1, n_blocks + 1))
        assert sum(t[3] for t in mine) == waves[w][2]
    assert st3["rays"] == sum(w[2] for w in waves)
    multi.close()
    one = api.MultiDeviceScene(s, [0])               # a single device: no merge at all
    got1, _ = one.render(p)
    assert np.array_equal(got1.view(np.uint32), want.view(np.uint32))
    one.close(); single.close()


def test_rccl_calls_of_the_merge_run_on_this_box(api):
    """yart_hip_multi_rccl_selftest: ncclCommInitAll + a grouped ncclSend / ncclRecv of a 4 MB slab (one rank, to itself) +
    ncclCommDestroy through the library — the transport calls of the multi-device merge, which otherwise need two GPUs."""
    L = api.lib()
    L.yart_hip_multi_rccl_selftest.argtypes = [ctypes.c_int, ctypes.c_uint32]
    rc = L.yart_hip_multi_rccl_selftest(0, 1 << 20)
    assert rc == 0, (rc, L.yart_hip_last_error())


# ---------------------------------------------------------------------------------------------------------------
# Two and more GPUs: these tests enable themselves the moment the box shows >= 2 devices (a one-GPU box skips them;
# everything around the transport runs there through the one-device rehearsals above).
# ---------------------------------------------------------------------------------------------------------------
def _n_devices():
    # (from the KFD topology in sysfs: evaluated at collection time, also by CPU-only runs — no HIP runtime call here)
    import bench
    return bench.visible_gpu_count()


@pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 HIP devices (RCCL branch of multi_device.inc)")
def test_multi_device_rccl_branch_equals_single_device(api):
    """MultiDeviceScene over DISTINCT devices: the grouped ncclSend / ncclRecv merge (multi_device.inc, `distinct`) must
    give the single-device frame bit for bit — whole tiles, 16-pixel blocks, several waves, every visible device."""
    from yart_amd import scenes
    n = min(_n_devices(), 8)
    s, p = scenes.sponza_class(240, 136, 8, 6, tex=128, sky=128)
    single = api.DeviceScene(s, device=0)
    want, st1 = single.render(p)
    for devs in ([0, 1], list(range(n))):
        multi = api.MultiDeviceScene(s, devs)
        got, st = multi.render(p)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), devs
        assert st["samples"] == st1["samples"] and st["rays"] == st1["rays"]
        q = dict(p, shard_tile=16, first_wave=2, max_wave=4)
        want2, _ = single.render(q)
        got2, _ = multi.render(q)
        assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32)), devs
        got3, _ = multi.render(q)                    # a second render re-uses the communicators
        assert np.array_equal(got3.view(np.uint32), want2.view(np.uint32)), devs
        multi.close()
    single.close()


def _nccl_rank(rank, world, port, q):
    import os as _os
    import sys as _sys
    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from tests.conftest import ROOT
    _sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from yart_amd import api as _api, dist as yd, scenes
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    s, p = scenes.sponza_class(240, 136, 8, 6, tex=128, sky=128)
    p = dict(p, shard_tile=16)
    ds = _api.DeviceScene(s, device=rank)
    fb = torch.zeros((136, 240, 4), dtype=torch.float32, device="cuda")
    yd.render_sharded(ds, p, fb, rank, world, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ok = True
    if rank == 0:
        want, _ = ds.render(p)
        ok = bool(np.array_equal(fb.cpu().numpy().view(np.uint32), want.view(np.uint32)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()
    ds.close()


@pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 HIP devices (process-per-GPU RCCL reduce)")
def test_two_process_render_sharded_nccl(api):
    """yart_amd.dist.render_sharded with backend nccl (= RCCL), one process per GPU: rank 0's merged frame equals the
    frame one device renders alone, bit for bit (what bench.py --gpus N times)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_nccl_rank, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert res[0] and res[1]


def test_path_pool_sizes(api):
    """YART_FLAG_PATH_POOL: the batch runs through a pool of path slots, a slot whose path has ended takes the batch's next
    (pixel, sample). Pools far smaller than the batch (many rounds of refill, the drain at the end of every batch), a pool of
    one wave, a pool larger than the batch, several batches per wave, two progressive waves: always the default pipeline's
    frame and ray count, bit for bit."""
    from yart_amd import scenes
    s, p = scenes.sponza_class(160, 90, 8, 8, tex=64, sky=64)
    scene = api.DeviceScene(s, device=0)
    ref, st0 = scene.render(p)
    for q in (dict(pool_paths=4096), dict(pool_paths=64), dict(pool_paths=1 << 22), dict(pool_paths=5000, max_batch_paths=30000),
              dict(pool_paths=3000, first_wave=2, max_wave=4)):
        base = ref
        if "first_wave" in q:
            base, _ = scene.render(dict(p, first_wave=2, max_wave=4))
        img, st = scene.render(dict(p, **q), flags=512)
        assert np.array_equal(img.view(np.uint32), base.view(np.uint32)), q
        assert int(st["rays"]) == int(st0["rays"]), q
        assert st["pipeline_flags"] & 512
    scene.close()


def test_bench_inproc_mode_rehearsal(api):
    """bench.py --inproc (one process, yart_hip_multi_render over N devices) on a small frame with both replicas on device 0
    (YART_BENCH_ONE_DEVICE: peer copies instead of RCCL): one JSON line, the frame's checksum equal to a single-device render's,
    and the per-bounce path counts of YartStats add up."""
    import sys
    from yart_amd import scenes
    env = dict(os.environ, YART_BENCH_ONE_DEVICE="1")
    args = ["--width", "96", "--height", "54", "--spp", "4", "--depth", "4", "--tex", "32", "--sky", "32", "--steps", "1", "--warmup", "0"]
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(GOLDEN), "..", "bench.py"), "--gpus", "2", "--inproc"] + args,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["mode"] == "inproc" and line["n_gpus"] == 2 and line["value"] > 0
    s, p = scenes.sponza_class(96, 54, 4, 4, tex=32, sky=32)
    scene = api.DeviceScene(s, device=0)
    img, st = scene.render(p)
    assert float(np.nan_to_num(img[..., :3]).sum(dtype=np.float64)) == line["frame_checksum"]
    assert int(st["rays"]) == line["rays_per_step"]
    pb = list(st["paths_at_bounce"])
    assert pb[0] == 96 * 54 * 4 and all(pb[i] >= pb[i + 1] for i in range(4)) and pb[1] > 0 and pb[5] == 0
    scene.close()


def test_concurrent_callers(api):
    """The C ABI is thread-safe per handle (include/yart_hip.h): host threads that render through ONE scene handle at the same time are
    serialised by the handle and each gets its own frame; threads on DIFFERENT handles of the same device run side by side. Every
    frame equals the one the same call returns alone."""
    import threading
    from yart_amd import scenes
    sa, pa = scenes.fuzz_case(3)
    sb, pb = scenes.fuzz_case(14)
    A, B = api.DeviceScene(sa, device=0), api.DeviceScene(sb, device=0)
    jobs = [(A, dict(pa, spp=4), 0), (A, dict(pa, spp=8), 512), (A, dict(pa, spp=16), 256), (B, dict(pb, spp=4), 0),
            (B, dict(pb, spp=8), 1), (A, dict(pa, spp=4), 4), (B, dict(pb, spp=16), 16), (A, dict(pa, spp=8), 0)]
    alone = [ds.render(p, flags=f)[0].copy() for ds, p, f in jobs]
    got, errors = [None] * len(jobs), []

    def work(k):
        try:
            for _ in range(3):
                ds, p, f = jobs[k]
                got[k] = ds.render(p, flags=f)[0].copy()
        except Exception as e:          # noqa: BLE001 (reported below, from the main thread)
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    for k in range(len(jobs)):
        assert np.array_equal(got[k].view(np.uint32), alone[k].view(np.uint32)), k
    A.close(); B.close()
