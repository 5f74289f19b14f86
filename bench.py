#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (W*H*spp / s) of the path-tracing hot path on the
Sponza-class scene at 1920x1080, 256 spp, 8 bounces (BASELINE.json configs[2]), tiles
sharded across N MI355X with an RCCL reduce of the framebuffer (configs[3] scheme).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete render of the frame (all 256 spp of every pixel, GMoN, blend,
and — for N > 1 — the reduce of the per-rank framebuffers to rank 0). Scene upload and
BVH build are outside the timed region; the framebuffer stays in HBM (a torch tensor).

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    # experiment knobs (the defaults ARE the BASELINE workload)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--tex", type=int, default=1024)
    ap.add_argument("--sky", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=8, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--flags", type=int, default=0, help="YART_FLAG_* pipeline variant (A/B experiments)")
    return ap.parse_args()


def cpu_baseline(scene, p, args):
    """Time the compiled reference (oracle/_ref/yart_ref, "reference") — or the CPU
    restatement (oracle/_build/yart_oracle, "port") if the reference binary is absent —
    on a bounded sample of the same workload: same scene, camera, resolution and bounce
    depth, `cpu_spp` samples per pixel, all host cores (64x64 tiles on a thread pool,
    the reference's own scheme)."""
    from yart_amd import scenes
    ref = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
    port = os.path.join(ROOT, "oracle", "_build", "yart_oracle")
    exe, kind = (ref, "reference") if os.path.exists(ref) else (port, "port")
    if not os.path.exists(exe):
        return None
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as td:
        sp, pp, out = os.path.join(td, "s.yscn"), os.path.join(td, "p.txt"), os.path.join(td, "o.f32")
        scene.save(sp)
        # the reference's own knobs only (the library's sharding / estimator keys mean nothing to it)
        q = {k: v for k, v in p.items() if k not in ("shard_tile", "estimator", "start_sample", "stop_sample")}
        scenes.write_params(pp, dict(q, spp=args.cpu_spp), threads=cores)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "render", sp, pp, out], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise SystemExit(f"bench.py: the CPU baseline ({exe}) failed: {r.stderr.strip()[-400:]}")
        info = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": round(info["msamples_per_s"], 4), "unit": "Msamples/s", "cores": cores, "kind": kind,
            "sample": f"same scene/camera, {p['size'][0]}x{p['size'][1]}, {args.cpu_spp} spp of {p['spp']}, "
                      f"depth {p['depth']}, {info['seconds']:.1f} s render ({wall:.1f} s incl. BVH build)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    from yart_amd import api, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # rehearsal knobs for a one-GPU box (never set by the driver): every rank on device 0, gloo instead of RCCL
    backend = os.environ.get("YART_BENCH_BACKEND", "nccl")
    if os.environ.get("YART_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    scene, p = scenes.sponza_class(args.width, args.height, args.spp, args.depth, tex=args.tex, sky=args.sky)
    W, H = p["size"]
    # N > 1: 16x16 pixel blocks are dealt to the ranks instead of whole 64x64 tiles — measured per-rank times of the
    # 8-way split on one GPU: 162-175 ms with tiles, 168-172 ms with 16x16 blocks (the slowest rank sets the step)
    shard = 16 if world > 1 else 0
    p = dict(p, shard_tile=shard)
    dscene = api.DeviceScene(scene, device=local_rank)
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    last = {}

    def step():
        st = dscene.render_into(fb, p, rank=rank, world_size=world, flags=args.flags, stream=stream)
        last.update(st)
        if world > 1:
            # non-owned tiles are exactly 0 on every rank -> the sum is the merged frame
            if backend == "nccl":
                dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM)
            else:
                dist.all_reduce(fb, op=dist.ReduceOp.SUM)      # gloo has no CUDA reduce

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, launches = 0.0, 0
    lean = (args.flags & 5) == 0       # default pipeline: the dominant kernel is the lean closest-hit kernel
    for _ in range(args.steps):
        step()
        if lean:
            kernel_ms += last["ms_extend_lean"]; launches += last["launches_extend_lean"]
        else:
            kernel_ms += last["ms_traverse"]; launches += last["launches_traverse"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_samples = W * H * p["spp"] * args.steps
    value = total_samples / dt * 1e-6
    out = {
        "metric": "Msamples/sec (W*H*spp/s), Sponza-class 1080p 8-bounce",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "sponza_class (generated atrium, %d triangles, env-lit) %dx%d, %d spp, %d bounces"
                               % (scene.n_triangles, W, H, p["spp"], p["depth"]),
                   "pipeline": "megakernel" if args.flags & 1 else "wavefront", "tiles": f"{shard or 64}x{shard or 64} pixel blocks, Morton order, round-robin over ranks",
                   "parallelism": f"tiles/{world}"},
        "rays_per_step": int(last.get("rays", 0)),
    }

    # ---- roofline of the dominant kernel (the path / traversal kernel) -----------------
    if not args.no_roofline and world == 1:
        # exact test counters from the instrumented twin library (one untimed pass; the
        # workload is deterministic so the counts are those of every timed pass)
        dscene.close()           # hand its batch buffers back: the counting pass is one batch too, launch for launch
        ds2 = api.DeviceScene(scene, device=local_rank, instrumented=True)
        _, st2 = ds2.render(p, flags=args.flags)
        ds2.close()
        shaded = st2.get("shaded_hits", 0)
        # B_traversal = 32*N_box + 52*N_tri + 48 per traversal. Default pipeline: the
        # dominant kernel is the lean closest-hit kernel k_wf_extend_fast (its own counters, its own
        # HIP-event time, one launch per bounce per batch); otherwise all traversal launches together.
        if lean:
            kname = "k_wf_extend_fast" if args.flags & 16 else "k_wf_extend_lean"
            trav_bytes = 32 * st2["lean_box_tests"] + 52 * st2["lean_tri_tests"] + 48 * st2["lean_traversals"]
            n_launch = max(1, last["launches_extend_lean"])
        else:
            kname = "k_render_mega" if args.flags & 1 else "closest-hit + shadow traversal kernels"
            trav_bytes = 32 * st2["box_tests"] + 52 * st2["tri_tests"] + 48 * st2["traversals"]
            n_launch = max(1, last["launches_traverse"])
        avg_ms = kernel_ms / max(1, launches)
        achieved = trav_bytes / n_launch / (avg_ms * 1e-3) * 1e-9
        out["stage_ms_per_step"] = {k: round(last[k], 2) for k in
                                    ("ms_extend", "ms_extend_lean", "ms_connect", "ms_shade", "ms_gmon", "ms_device")}
        traffic = None
        if lean and (W, H, p["spp"], p["depth"], args.tex, args.sky) == (1920, 1080, 256, 8, 1024, 2048):
            # memory-side bytes per launch of this kernel from the committed PMC passes (tools/pmc_hbm.sh)
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_hbm.json")))
                k = next(v for n, v in prof["kernels"].items() if n.startswith(kname))
                traffic = k["read_bytes_per_launch"] + k["write_bytes_per_launch"]
            except (OSError, StopIteration, KeyError, ValueError):
                traffic = None
        out["roofline"] = {
            "bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
            "avg_launch_ms": round(avg_ms, 3), "launches_per_step": n_launch,
            "algorithmic_bytes_per_launch": int(trav_bytes / n_launch),
            "counts_per_step": {"traversals": st2["traversals"], "box_tests": st2["box_tests"],
                                "tri_tests": st2["tri_tests"], "shaded_hits": shaded,
                                "lean_traversals": st2["lean_traversals"], "lean_box_tests": st2["lean_box_tests"],
                                "lean_tri_tests": st2["lean_tri_tests"]},
        }
    if not args.no_cpu_baseline and world == 1:
        cb = cpu_baseline(scene, p, args)
        if cb:
            out["cpu_baseline"] = cb
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
