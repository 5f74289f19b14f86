"""N > 1 path on CPU: two gloo ranks each hold the pixels of their own tiles (zeros
elsewhere); the merge (reduce SUM to rank 0) must give back the full reference frame bit
for bit, and the tile ownership must be a partition of the image."""
import os
import sys

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT


def test_tile_ownership_is_a_partition():
    from yart_amd import dist as yd
    for (w, h, tile, world) in [(1920, 1080, 64, 8), (128, 128, 64, 2), (100, 70, 32, 3), (64, 64, 64, 4)]:
        total = sum(yd.pixel_mask(w, h, tile, r, world).astype(np.int32) for r in range(world))
        assert np.all(total == 1)
        counts = [int((yd.tile_owners(w, h, tile, world) == r).sum()) for r in range(world)]
        assert max(counts) - min(counts) <= 1      # round-robin balance


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from yart_amd import dist as yd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = np.fromfile(os.path.join(GOLDEN, "cornell.f32"), np.float32).reshape(128, 128, 4)
    mask = yd.pixel_mask(128, 128, 64, rank, world)
    part = torch.from_numpy(np.where(mask[..., None], ref, 0.0).astype(np.float32))
    yd.merge(part, 0)
    ok = bool(np.array_equal(part.numpy().view(np.uint32), ref.view(np.uint32))) if rank == 0 else True
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] and res[1]


def test_bench_dry_launch_spawns_the_ranks_it_was_asked_for():
    """`python bench.py --gpus 2 --dry-launch` — the command form the driver uses, without torchrun around it — starts two
    gloo ranks itself, each holds the blocks the library's partition gives it of a synthetic frame, the reduce gives the
    full frame back on rank 0, and the line says n_gpus = 2 (--gpus was parsed and never read)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dry_launch"] and line["n_gpus"] == 2 and line["merged_equals_full_frame"]
    assert len(line["pixels_per_rank"]) == 2 and sum(line["pixels_per_rank"]) == line["frame"][0] * line["frame"][1]


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """No GPU here: `--gpus 2` must fail loudly (exit 2) instead of timing one device, and a torchrun world that
    differs from --gpus is refused as well."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    sys.path.insert(0, ROOT)
    import bench
    if bench.visible_gpu_count() < 2:             # (counted from sysfs: this process does not open the GPU)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 2 and "refusing" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr


def test_bench_inproc_refuses_without_the_devices():
    """--inproc (one process, yart_hip_multi_render over N devices) refuses to run on fewer devices than asked."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    if bench.visible_gpu_count() >= 2:
        pytest.skip("two devices are visible: the refusal cannot be provoked here")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "YART_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--inproc"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "--inproc" in r.stderr


def test_bench_cpu_baseline_leg_runs_on_a_small_case():
    """bench.py's cpu_baseline leg (the compiled reference, or the oracle, timed on the host cores) on a 32x32 case,
    with the library-only keys the bench carries in its parameter dict: the timing entry, the frame the parity gate
    compares the GPU render with, and the oracle's test counters in the reference's traversal order."""
    import argparse
    import numpy as np
    import bench
    from yart_amd import scenes
    scene, p = scenes.cornell(32, 32, 2, 3)
    res, frame = bench.cpu_baseline(scene, dict(p, shard_tile=16, estimator=0, max_batch_paths=0),
                                    argparse.Namespace(cpu_spp=1, ref_order_spp=1))
    if res is None:
        import pytest
        pytest.skip("neither oracle/_ref/yart_ref nor oracle/_build/yart_oracle is built")
    cb, ref_order = res
    assert cb["unit"] == "Msamples/s" and cb["value"] > 0 and cb["kind"] in ("reference", "port") and cb["cores"] >= 1
    assert frame.shape == (32, 32, 4) and np.all(frame[..., 3] == 1.0)
    if ref_order is not None:
        assert ref_order["traversals"] > ref_order["shadow_traversals"] > 0
        assert ref_order["box_tests"] > ref_order["shadow_box_tests"] > 0 and ref_order["tri_tests"] > 0


def test_bench_kernel_name_parsing():
    import bench
    assert bench.kernel_base("void (anonymous namespace)::k_wf_shade<false>((anonymous namespace)::WfArgs)") == "k_wf_shade"
    assert bench.kernel_base("(anonymous namespace)::k_wf_post((anonymous namespace)::WfArgs)") == "k_wf_post"
    assert bench.kernel_base("__amd_rocclr_copyBuffer") is None
