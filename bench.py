#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (W*H*spp / s) of the path-tracing hot path on the
Sponza-class scene at 1920x1080, 256 spp, 8 bounces (BASELINE.json configs[2]), tiles
sharded across N MI355X with an RCCL reduce of the framebuffer (configs[3] scheme).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` is what decides the number of ranks. Started WITHOUT torchrun's environment and N > 1, this process
launches the N ranks itself (`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, BEFORE
it has touched the GPU), relays rank 0's JSON line and exits with the child's code; it refuses (exit 2) when fewer
than N devices are visible. Started BY torchrun, WORLD_SIZE must equal --gpus (exit 2 otherwise): `n_gpus` in the
line is the world size RCCL actually ran with, never a number taken from the command line alone.
`--dry-launch` rehearses exactly that launch path without a GPU: N gloo ranks reduce synthetic frames cut by the
library's tile partition (tests/test_dist_cpu.py).

A "step" is one complete render of the frame (all 256 spp of every pixel, GMoN, blend,
and — for N > 1 — the reduce of the per-rank framebuffers to rank 0). Scene upload and
BVH build are outside the timed region; the framebuffer stays in HBM (a torch tensor).

Rank 0 prints ONE JSON line (DESIGN.md §5 explains every field). The line also carries (at N > 1 for rank 0's share of the frame,
with `per_rank` = every rank's device time and stage times and `reduce_ms` = the RCCL reduce per step)
  parity        the same scene / camera / depth at `--cpu-spp` samples rendered on the GPU and compared with the
                frame the CPU baseline leg just rendered (RMSE in linear HDR, identical-pixel fraction); the bench
                exits with status 3 if RMSE >= 1e-3 (every timed run is checked)
  roofline      the kernel with the largest share of the step, `rooflines` all three large kernels: algorithmic
                bytes per launch / HIP-event launch time against the roof that binds (HBM for the shade kernel,
                the L2 for the BVH walks whose working set is cache resident), and `traffic` = FETCH_SIZE x 2 +
                WRITE_SIZE of that kernel measured by two rocprofv3 --pmc child runs of this workload started
                BEFORE this process touches the GPU (null when they cannot run); a third child run collects the SQ
                counters of the same workload: per kernel `lane_util` (how full the issued vector instructions are),
                `valu_busy` (share of the SIMD cycles that issue one) and, for the traversal kernels, `valu_frac` (the
                box + triangle tests' lane-operations against the peak lane-operation rate)
  cpu_baseline  the compiled reference on the host cores, bounded sample.
  one_batch     (N = 1) the headline's frame rendered as ONE batch (max_batch_paths = its path count, 152 GB of path state): how
                rounds 1-4 timed the headline. `value` itself is measured at the library's default (2^28 paths per batch).
  secondary     (N = 1) BASELINE.json's other configurations at the library's defaults, 1-2 steps each, each with its own parity
                gate against the compiled reference (a failing gate -> exit status 3) and the shade kernel's roofline:
                configs[4] McLaren-class 3840x2160 x 512 spp, configs[3]'s per-GPU workload Sponza-class 1080p x 1024 spp,
                configs[1] Cornell 512x512 x 64 spp through the megakernel. `--no-secondary` skips them.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md §HBM
L2_PEAK_GBPS = 34500.0          # aggregate L2 (8 XCDs x 4 MiB), same guide §L2
LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9   # vector lane-operations per second: 256 CUs x 4 SIMDs x 16 lanes per clock at 2.4 GHz
# the useful lane-instructions of one test as the lean kernels code it (ISA counts): the reference's slab test of one box
# (6 mul, 6 add, 2 max3, 2 min3, compare, select = 18) and its Moeller-Trumbore test of one triangle (2 cross, 4 dot, divide, ~45)
BOX_TEST_LANE_OPS, TRI_TEST_LANE_OPS = 18, 45
# waves per SIMD the kernels are built for (csrc/wavefront_kernels.inc: YART_LEAN_WAVES, YART_SHADE_WAVES): VALU-issue share of a
# SIMD's cycles = valu_active_per_wave_cycle x resident waves
KERNEL_WAVES_PER_SIMD = {"k_wf_extend_lean": 7, "k_wf_shadow_lean": 7, "k_wf_shade": 3}
RMSE_TOL = 1e-3                 # BASELINE.json north_star


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
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs (roofline.traffic = null)")
    ap.add_argument("--cpu-spp", type=int, default=8, help="spp of the bounded CPU-baseline / parity sample")
    ap.add_argument("--ref-order-spp", type=int, default=1, help="spp of the oracle run that counts box / triangle tests in the reference's traversal order")
    ap.add_argument("--flags", type=int, default=0, help="YART_FLAG_* pipeline variant (A/B experiments)")
    ap.add_argument("--batch-paths", type=int, default=0,
                    help="YartRenderParams::max_batch_paths: 0 (default) = the library's default (2^28 paths per batch: what a caller gets), "
                         "-1 = the whole frame as one batch (how rounds 1-4 timed the headline)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (BASELINE configs[4], [3], [1]) and the one-batch extra of the headline")
    ap.add_argument("--inproc", action="store_true",
                    help="time the in-process form instead: ONE process, yart_hip_multi_render over --gpus devices (a host thread per "
                         "device, the devices' own pixels merged on device 0 with RCCL send / recv: 1/N of the frame's bytes)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # one un-timed step, no torch (run under rocprofv3 --pmc)
    ap.add_argument("--dry-launch", action="store_true",
                    help="no GPU: start the --gpus N ranks with gloo and reduce synthetic frames (launch-path rehearsal)")
    ap.add_argument("--keep-pmc", default="", help="directory to keep the raw counter csv files in (e.g. profiles/...)")
    return ap.parse_args()


def workload(args):
    """The scene and the render parameters. `--batch-paths` sets YartRenderParams::max_batch_paths: 0 (default since round 5) = the
    library's default, a fixed 2^28 paths per batch (77 GB; the 1080p x 256 spp frame then takes two batches); -1 = the frame's own
    path count, i.e. ONE batch (how rounds 1-4 timed the headline: 152 GB of path state, ~10 ms less per step) — the line carries
    that figure as the named extra `one_batch`. The frame does not depend on it."""
    from yart_amd import scenes
    scene, p = scenes.sponza_class(args.width, args.height, args.spp, args.depth, tex=args.tex, sky=args.sky)
    frame = args.width * args.height * args.spp
    p["max_batch_paths"] = min(frame, (1 << 31) - 64) if args.batch_paths < 0 else args.batch_paths
    return scene, p


# ---------------------------------------------------------------------------------------------------
# memory-side traffic per kernel: rocprofv3 --pmc child runs (separate passes for FETCH_SIZE and WRITE_SIZE)
# ---------------------------------------------------------------------------------------------------
def under_profiler():
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "") + os.environ.get("HSA_TOOLS_LIB", "")
    return "rocprof" in pre


def kernel_base(name):
    i = name.find("k_")
    if i < 0:
        return None
    k = name[i:]
    for stop in ("<", "("):
        j = k.find(stop)
        if j >= 0:
            k = k[:j]
    return k


def pmc_pass(args, counters, keep_name=None):
    """One rocprofv3 --pmc child run of ONE un-timed step of the workload: {kernel: {counter: sum over its launches, "launches": n}},
    or None (no profiler, the pass failed or did not finish)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    keep = os.path.join(ROOT, args.keep_pmc) if args.keep_pmc else None
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        env = dict(os.environ, TMPDIR="/tmp")
        d = os.path.join(td, "pass")
        cmd = [exe, "--pmc"] + list(counters) + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--",
               sys.executable or "python3", os.path.join(ROOT, "bench.py"), "--pmc-child",
               "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp), "--depth", str(args.depth),
               "--tex", str(args.tex), "--sky", str(args.sky), "--flags", str(args.flags), "--batch-paths", str(args.batch_paths)]
        # its own process group: should the pass hang, the profiler AND the python under it are ended together
        try:
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                    start_new_session=True)
        except OSError:
            return None
        try:
            so, se = proc.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.wait()
            sys.stderr.write(f"bench.py: rocprofv3 --pmc {' '.join(counters)} pass did not finish in 240 s\n")
            return None
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if proc.returncode != 0 or not files:
            sys.stderr.write(f"bench.py: rocprofv3 --pmc {' '.join(counters)} pass failed ({proc.returncode}): {se.strip()[-300:]}\n")
            return None
        if keep:
            os.makedirs(keep, exist_ok=True)
            shutil.copy(files[0], os.path.join(keep, f"pmc_{(keep_name or counters[0]).lower()}_counter_collection.csv"))
        out = {}
        seen = {}
        for row in csv.DictReader(open(files[0])):
            k = kernel_base(row["Kernel_Name"])
            if k is None or row["Counter_Name"] not in counters:
                continue
            e = out.setdefault(k, {"launches": 0})
            e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            if (k, row["Dispatch_Id"]) not in seen:
                seen[(k, row["Dispatch_Id"])] = True
                e["launches"] += 1
        return out


def pmc_traffic(args):
    """{kernel: {launches, read_bytes, write_bytes}} summed over ONE step of the workload, or None.
    FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side, collected in separate passes; FETCH_SIZE is doubled (gfx950 tallies
    128-B read requests at 64 B) exactly as MI355X_MICROARCH.md §HBM prescribes; Infinity-Cache hits are included."""
    out = {}
    for counter, field, scale in (("FETCH_SIZE", "read_bytes", 2048.0), ("WRITE_SIZE", "write_bytes", 1024.0)):
        r = pmc_pass(args, [counter])
        if r is None:
            return None
        for k, v in r.items():
            e = out.setdefault(k, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
            e[field] += v.get(counter, 0.0) * scale
            if counter == "FETCH_SIZE":
                e["launches"] += v["launches"]
    return out


SQ_COUNTERS = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY"]


def pmc_efficiency(args):
    """{kernel: {lane_util, valu_active_per_wave_cycle, ...}} of ONE step of the timed workload (a third counter pass): how full the
    64 lanes of the issued vector instructions are, and the share of a resident wave's cycles in which it issues one — the two figures
    that the bandwidth fractions hide (a kernel at 0.4 of the L2 roof with half of its lanes idle)."""
    r = pmc_pass(args, SQ_COUNTERS, keep_name="sq")
    if r is None:
        return None
    out = {}
    for k, v in r.items():
        act, wc = v.get("SQ_ACTIVE_INST_VALU", 0.0), v.get("SQ_WAVE_CYCLES", 0.0)
        if act <= 0 or wc <= 0:
            continue
        out[k] = {"lane_util": round(v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (act * 64.0), 4),
                  "valu_active_per_wave_cycle": round(act / wc, 4),
                  "wait_inst_per_wave_cycle": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                  "valu_wave_instructions": int(v.get("SQ_INSTS_VALU", 0.0)), "salu_wave_instructions": int(v.get("SQ_INSTS_SALU", 0.0)),
                  "launches": v["launches"]}
    return out


def pmc_child(args):
    """One un-timed step through the host-buffer entry point (no torch), for the counter passes."""
    from yart_amd import api
    scene, p = workload(args)
    ds = api.DeviceScene(scene, device=0)
    ds.render(p, flags=args.flags)
    ds.close()


# ---------------------------------------------------------------------------------------------------
# CPU legs: the compiled reference (timing + the parity frame) and the oracle (reference-order test counts)
# ---------------------------------------------------------------------------------------------------
def cpu_baseline(scene, p, args):
    """Time the compiled reference (oracle/_ref/yart_ref, "reference") — or the CPU restatement
    (oracle/_build/yart_oracle, "port") if the reference binary is absent — on a bounded sample of the same
    workload: same scene, camera, resolution and bounce depth, `cpu_spp` samples per pixel, all host cores
    (64x64 tiles on a thread pool, the reference's own scheme). Returns (json entry, frame)."""
    import numpy as np
    from yart_amd import scenes
    ref = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
    port = os.path.join(ROOT, "oracle", "_build", "yart_oracle")
    exe, kind = (ref, "reference") if os.path.exists(ref) else (port, "port")
    if not os.path.exists(exe):
        return None, None
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as td:
        sp, pp, out = os.path.join(td, "s.yscn"), os.path.join(td, "p.txt"), os.path.join(td, "o.f32")
        scene.save(sp)
        # the reference's own knobs only (the library's sharding / estimator keys mean nothing to it)
        q = {k: v for k, v in p.items() if k not in ("shard_tile", "estimator", "start_sample", "stop_sample", "max_batch_paths")}
        scenes.write_params(pp, dict(q, spp=args.cpu_spp), threads=cores)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "render", sp, pp, out], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise SystemExit(f"bench.py: the CPU baseline ({exe}) failed: {r.stderr.strip()[-400:]}")
        info = json.loads(r.stdout.strip().splitlines()[-1])
        frame = np.fromfile(out, np.float32).reshape(p["size"][1], p["size"][0], 4)
        ref_order = None
        if os.path.exists(port) and args.ref_order_spp > 0:
            # box / triangle tests per ray in the REFERENCE's traversal order (oracle counters)
            scenes.write_params(pp, dict(q, spp=args.ref_order_spp), threads=cores)
            r2 = subprocess.run([port, "render", sp, pp, out], capture_output=True, text=True)
            if r2.returncode == 0:
                ref_order = json.loads(r2.stdout.strip().splitlines()[-1])
    entry = {"value": round(info["msamples_per_s"], 4), "unit": "Msamples/s", "cores": cores, "kind": kind,
             "sample": f"same scene/camera, {p['size'][0]}x{p['size'][1]}, {args.cpu_spp} spp of {p['spp']}, "
                       f"depth {p['depth']}, {info['seconds']:.1f} s render ({wall:.1f} s incl. BVH build)"}
    return (entry, ref_order), frame


def reference_frame(scene, p, spp):
    """(frame, info, kind) of the compiled reference (or the CPU restatement where it is not built) at `spp` samples per pixel on all
    host cores — the checker of a secondary workload's parity gate."""
    import numpy as np
    from yart_amd import scenes
    ref = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
    port = os.path.join(ROOT, "oracle", "_build", "yart_oracle")
    exe, kind = (ref, "reference") if os.path.exists(ref) else (port, "port")
    if not os.path.exists(exe):
        return None, None, None
    with tempfile.TemporaryDirectory() as td:
        sp, pp, out = os.path.join(td, "s.yscn"), os.path.join(td, "p.txt"), os.path.join(td, "o.f32")
        scene.save(sp)
        q = {k: v for k, v in p.items() if k not in ("shard_tile", "estimator", "start_sample", "stop_sample", "max_batch_paths")}
        scenes.write_params(pp, dict(q, spp=spp), threads=os.cpu_count() or 1)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "render", sp, pp, out], capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise SystemExit(f"bench.py: the CPU checker ({exe}) failed: {r.stderr.strip()[-400:]}")
        info = json.loads(r.stdout.strip().splitlines()[-1])
        info["wall_s"] = wall
        return np.fromfile(out, np.float32).reshape(p["size"][1], p["size"][0], 4), info, kind


def shade_roofline(c, ms_sum, launches, steps):
    """The shade kernel's roofline entry from the exact counters `c` of one instrumented pass (SURVEY §8(d) B_shade) and its summed
    HIP-event time over `steps` timed steps."""
    n_l = max(1, launches // max(1, steps))
    t = (ms_sum / max(1, launches)) * 1e-3
    b = 116 * c["shaded_hits"] + 64 * c["shade_entries"] + c["texture_tap_bytes"]
    e = {"kernel": "k_wf_shade", "bound": "hbm", "peak": HBM_PEAK_GBPS, "avg_launch_ms": round(t * 1e3, 3), "launches_per_step": n_l,
         "algorithmic_bytes_per_launch": int(b / n_l),
         "model": "116 B x shaded hits + 64 B x entries + 4 taps x channels x texel bytes per lookup (SURVEY §8(d) B_shade)",
         "units_per_step": {"shaded_hits": c["shaded_hits"], "entries": c["shade_entries"], "texture_tap_bytes": c["texture_tap_bytes"]}}
    e["achieved"] = round(e["algorithmic_bytes_per_launch"] / t * 1e-9, 2) if t > 0 else 0.0
    e["unit"] = "GB/s"
    e["frac"] = round(e["achieved"] / e["peak"], 5)
    e["traffic"] = None          # (counter passes are run for the headline workload only)
    return e


def secondary_workloads(args, api, torch, np, headline_scene, headline_p, device):
    """BASELINE.json's other configurations, timed by the driver's own run of this file: each at the LIBRARY's
    defaults (max_batch_paths = 0 -> 2^28 paths per batch), each with its own parity gate — the same scene / camera / depth at a
    bounded sample count on the GPU against the compiled reference's frame — and, for the wavefront pipeline, the shade kernel's
    roofline from an instrumented pass of the full workload. Returns (list of entries, all gates ok)."""
    from yart_amd import scenes
    out, ok = [], True
    specs = [
        # (name, BASELINE config, scene factory, pipeline flags, warm-up, steps, parity spp)
        ("mclaren_class", "configs[4]", lambda: scenes.mclaren_class(3840, 2160, 512, 8, detail=1.0), 0, 1, 2, 1),
        ("sponza_class", "configs[3]", lambda: (headline_scene, dict(headline_p, spp=1024)), 0, 1, 2, 4),      # the headline's scene and camera at 1024 spp
        ("cornell", "configs[1]", lambda: scenes.cornell(512, 512, 64, 4), 1, 2, 5, 64),
    ]
    stream = torch.cuda.current_stream().cuda_stream
    for name, cfg, make, flags, warm, steps, pspp in specs:
        t_make = time.perf_counter()
        scene, p = make()
        W, H = p["size"]
        p = dict(p, max_batch_paths=0)
        ds = api.DeviceScene(scene, device=device)
        t_make = time.perf_counter() - t_make
        fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        for _ in range(warm):
            ds.render_into(fb, p, flags=flags, stream=stream)
        torch.cuda.synchronize()
        ms_shade = 0.0; n_shade = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            st = ds.render_into(fb, p, flags=flags, stream=stream)
            ms_shade += st["ms_shade_kernel"]; n_shade += st["launches_shade_kernel"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        e = {"workload": "%s (%d triangles) %dx%d, %d spp, %d bounces" % (name, scene.n_triangles, W, H, p["spp"], p["depth"]),
             "baseline_config": cfg, "pipeline": "megakernel" if flags & 1 else "wavefront", "pipeline_flags": int(st["pipeline_flags"]),
             "max_batch_paths": 0, "value": round(W * H * p["spp"] * steps / dt * 1e-6, 3), "unit": "Msamples/s", "steps": steps, "warmup": warm,
             "ms_per_step": round(dt / steps * 1e3, 3), "rays_per_step": int(st["rays"]),
             "paths_at_bounce": [int(x) for x in st.get("paths_at_bounce", [])][: p["depth"] + 1],
             "scene_build_s": round(t_make, 2)}
        # parity gate: `pspp` samples per pixel of the whole frame, GPU against the compiled reference
        ref, info, kind = reference_frame(scene, p, pspp)
        if ref is not None:
            img, _ = ds.render(dict(p, spp=pspp), flags=flags)
            err = float(np.sqrt(np.mean((np.nan_to_num(img[..., :3]).astype(np.float64) - np.nan_to_num(ref[..., :3])) ** 2)))
            same = float(np.mean(np.all(img.view(np.uint32) == ref.view(np.uint32), axis=-1)))
            e["parity"] = {"rmse": err, "identical_pixel_fraction": round(same, 6), "tolerance": RMSE_TOL, "ok": bool(err < RMSE_TOL),
                           "against": kind, "sample": f"{W}x{H}, {pspp} spp, depth {p['depth']} ({info['wall_s']:.1f} s on {os.cpu_count()} host threads)"}
            e["cpu_baseline"] = {"value": round(info["msamples_per_s"], 4), "unit": "Msamples/s", "cores": os.cpu_count() or 1, "kind": kind}
            ok = ok and err < RMSE_TOL
        ds.close()
        del fb
        if not (flags & 1) and not args.no_roofline:
            ds2 = api.DeviceScene(scene, device=device, instrumented=True)
            _, c = ds2.render(p, flags=flags)
            ds2.close()
            e["rooflines"] = [shade_roofline(c, ms_shade, n_shade, steps)]
        elif flags & 1:
            e["rooflines"] = None
            e["note"] = "one kernel (k_render_mega) walks, shades and connects: there is no separate shade kernel to put a roofline on"
        out.append(e)
    return out, ok


def visible_gpu_count():
    """GPUs this process could open, WITHOUT touching the HIP runtime (torch.cuda.device_count() falls back to hipGetDeviceCount on
    ROCm, which initialises HSA / KFD in the caller and keeps the handle open): the KFD topology's nodes with SIMDs, cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one of them is set."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
    except OSError:
        return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def inproc_run(args):
    """--inproc: the frame rendered by ONE process on --gpus devices through yart_hip_multi_render (csrc/multi_device.inc: one host
    thread per device, pixel blocks dealt round-robin, every device's own pixels sent to device 0 with grouped RCCL send / recv —
    1/N of the frame per device instead of the whole-frame reduce(SUM) of the process-per-GPU form) — the same workload, warm-up
    and step count; the frame is delivered to a host buffer each step, as the C ABI's blocking entry does. Prints one JSON line."""
    import numpy as np
    from yart_amd import api
    n = args.gpus
    one = bool(os.environ.get("YART_BENCH_ONE_DEVICE"))          # rehearsal on a one-GPU box: every replica on device 0 (peer copies, no RCCL)
    if not one and visible_gpu_count() < n:
        sys.stderr.write(f"bench.py: --inproc --gpus {n} but only {visible_gpu_count()} HIP device(s) are visible\n")
        return 2
    scene, p = workload(args)
    p = dict(p, shard_tile=16 if n > 1 else 0)
    ms = api.MultiDeviceScene(scene, [0] * n if one else list(range(n)))
    for _ in range(args.warmup):
        ms.render(p, flags=args.flags)
    t0 = time.perf_counter()
    st = {}
    for _ in range(args.steps):
        img, st = ms.render(p, flags=args.flags)
    dt = time.perf_counter() - t0
    W, H = p["size"]
    out = {"metric": "Msamples/sec (W*H*spp/s), Sponza-class 1080p 8-bounce", "value": round(W * H * p["spp"] * args.steps / dt * 1e-6, 3),
           "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "mode": "inproc",
           "config": {"workload": "sponza_class (generated atrium, %d triangles, env-lit) %dx%d, %d spp, %d bounces" % (scene.n_triangles, W, H, p["spp"], p["depth"]),
                      "parallelism": f"tiles/{n}: one process, yart_hip_multi_render ({'every replica on device 0: rehearsal' if one else 'one device per replica'}), "
                                     "slab merge on device 0, frame copied to the host every step"},
           "rays_per_step": int(st.get("rays", 0)), "frame_checksum": float(np.nan_to_num(img[..., :3]).sum(dtype=np.float64))}
    print(json.dumps(out), flush=True)
    return 0


def self_launch(args):
    """--gpus N > 1 (or --dry-launch) without torchrun's environment: start the N ranks as children of THIS process,
    which has not touched the GPU, relay their output (rank 0 prints the JSON line) and return the launcher's code."""
    import socket
    n = args.gpus
    if n < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if not args.dry_launch and not os.environ.get("YART_BENCH_ONE_DEVICE"):
        have = visible_gpu_count()                # (sysfs: the launcher never opens the GPU)
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} HIP device(s) are visible: refusing to measure fewer GPUs than asked\n")
            return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable or "python3", "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_rank(args, rank, world):
    """One rank of --dry-launch: gloo on the CPU, a synthetic frame (a function of the pixel) cut by the library's
    block partition, the same reduce the GPU path does, checked on rank 0. No GPU, no library call."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from yart_amd import dist as yd
    dist.init_process_group("gloo")
    W, H = min(args.width, 256), min(args.height, 144)
    ys, xs = np.mgrid[0:H, 0:W]
    full = np.stack([xs * 0.25 + 1.0, ys * 0.5 + 2.0, (xs ^ ys) * 1.0, np.ones_like(xs, dtype=np.float64)], -1).astype(np.float32)
    block = 16 if world > 1 else 64
    mask = yd.pixel_mask(W, H, block, rank, world)
    fb = torch.from_numpy(np.where(mask[..., None], full, 0.0).astype(np.float32))
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        part = fb.clone()
        yd.merge(part, 0)
    dt = time.perf_counter() - t0
    counts = [None] * world
    dist.all_gather_object(counts, int(mask.sum()))
    status = 0
    if rank == 0:
        ok = bool(np.array_equal(part.numpy().view(np.uint32), full.view(np.uint32)))
        print(json.dumps({"dry_launch": True, "n_gpus": dist.get_world_size(), "backend": "gloo", "steps": args.steps,
                          "frame": [W, H], "block": block, "pixels_per_rank": counts, "merged_equals_full_frame": ok,
                          "reduce_ms": round(dt / max(1, args.steps) * 1e3, 3), "data": "synthetic frames, no GPU"}), flush=True)
        status = 0 if ok and sum(counts) == W * H else 4
    dist.barrier()
    dist.destroy_process_group()
    return status


STAGE_KEYS = ("ms_device", "ms_extend", "ms_extend_lean", "ms_connect", "ms_shadow_lean", "ms_shade", "ms_shade_kernel", "ms_gmon")


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    if args.inproc:
        return inproc_run(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus != 1 or args.dry_launch):
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a mismatched n_gpus\n")
        return 2
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_launch:
        return dry_rank(args, rank, world)

    # counter passes first: child processes, started before this process has initialised the GPU. At N > 1 they are
    # left out (a profiler child on GPU 0 while seven other ranks hold their devices is not a combination to try on
    # a shared node): `traffic` is then null and the N = 1 line / profiles/ carry the memory-side bytes.
    pmc = eff = None
    if world == 1 and not args.no_roofline and not args.no_pmc and not under_profiler():
        pmc = pmc_traffic(args)
        eff = pmc_efficiency(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from yart_amd import api

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # rehearsal knobs for a one-GPU box (never set by the driver): every rank on device 0, gloo instead of RCCL
    backend = os.environ.get("YART_BENCH_BACKEND", "nccl")
    if os.environ.get("YART_BENCH_ONE_DEVICE"):
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        sys.stderr.write(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)\n")
        return 2
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != world:
            raise SystemExit("bench.py: the process group's world size differs from WORLD_SIZE")

    scene, p = workload(args)
    W, H = p["size"]
    # N > 1: 16x16 pixel blocks are dealt to the ranks instead of whole 64x64 tiles — measured per-rank times of the
    # 8-way split on one GPU: 162-175 ms with tiles, 168-172 ms with 16x16 blocks (the slowest rank sets the step)
    shard = 16 if world > 1 else 0
    p = dict(p, shard_tile=shard)
    dscene = api.DeviceScene(scene, device=local_rank)
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    last = {}
    reduce_ev = []

    def merge_frame(t):
        # non-owned blocks are exactly 0 on every rank -> the sum is the merged frame
        if backend == "nccl":
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)      # gloo has no CUDA reduce

    def step(timed=False):
        st = dscene.render_into(fb, p, rank=rank, world_size=world, flags=args.flags, stream=stream)
        last.update(st)
        if world > 1:
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                merge_frame(fb)
                b.record()
                reduce_ev.append((a, b))
            else:
                merge_frame(fb)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    # per-kernel HIP-event time on the render stream (YartStats), summed over the timed steps
    kern = {"k_wf_shade": [0.0, 0], "k_wf_extend_lean": [0.0, 0], "k_wf_shadow_lean": [0.0, 0]}
    stage_sum = {k: 0.0 for k in STAGE_KEYS}
    for _ in range(args.steps):
        step(timed=True)
        for k, (ms, n) in (("k_wf_shade", ("ms_shade_kernel", "launches_shade_kernel")),
                           ("k_wf_extend_lean", ("ms_extend_lean", "launches_extend_lean")),
                           ("k_wf_shadow_lean", ("ms_shadow_lean", "launches_shadow_lean"))):
            kern[k][0] += last[ms]; kern[k][1] += last[n]
        for k in STAGE_KEYS:
            stage_sum[k] += last[k]
    fence()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank's own device time and stage times (per step) and the time its stream spent in the reduce
        mine = {k: round(stage_sum[k] / max(1, args.steps), 3) for k in STAGE_KEYS}
        mine["reduce_ms"] = round(sum(a.elapsed_time(b) for a, b in reduce_ev) / max(1, len(reduce_ev)), 3)
        mine["rank"] = rank
        mine["samples"] = int(last.get("samples", 0))
        mine["paths_at_bounce"] = [int(x) for x in last.get("paths_at_bounce", [])][: p["depth"] + 1]   # this rank's live paths per bounce
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    total_samples = W * H * p["spp"] * args.steps
    value = total_samples / dt * 1e-6
    n_ranks = dist.get_world_size() if world > 1 else 1
    out = {
        "metric": "Msamples/sec (W*H*spp/s), Sponza-class 1080p 8-bounce",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n_ranks, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "sponza_class (generated atrium, %d triangles, env-lit) %dx%d, %d spp, %d bounces"
                               % (scene.n_triangles, W, H, p["spp"], p["depth"]),
                   "pipeline": "megakernel" if args.flags & 1 else "wavefront", "tiles": f"{shard or 64}x{shard or 64} pixel blocks, Morton order, round-robin over ranks",
                   "parallelism": f"tiles/{n_ranks}: one process per GPU (torch.distributed, backend {backend if world > 1 else 'none'}), "
                                  "blocks sharded by rank inside the library, one reduce(SUM) of the frame to rank 0 per step",
                   "pipeline_flags": int(last.get("pipeline_flags", args.flags)),
                   "max_batch_paths": int(p.get("max_batch_paths", 0)),
                   "batches": "the frame is one batch (max_batch_paths = its path count: 152 GB of path state)" if p.get("max_batch_paths", 0) >= W * H * p["spp"]
                              else "the library's default (max_batch_paths = 0: 2^28 paths per batch, what a caller gets; `one_batch` = the same "
                                   "frame as ONE batch, how rounds 1-4 timed it)" if not p.get("max_batch_paths", 0)
                              else "max_batch_paths = %d" % int(p.get("max_batch_paths", 0))},
        "rays_per_step": int(last.get("rays", 0)),
        "paths_at_bounce": [int(x) for x in last.get("paths_at_bounce", [])][: p["depth"] + 1],     # rank 0's live paths entering each bounce
    }
    if per_rank is not None:
        slow = max(per_rank, key=lambda r: r["ms_device"])
        ideal = sum(r["ms_device"] for r in per_rank) / len(per_rank)
        out["rays_per_step"] = None          # (rank 0's share only is known here; see per_rank samples)
        out["per_rank"] = per_rank
        out["slowest_rank"] = {"rank": slow["rank"], "stage_ms_per_step": {k: slow[k] for k in STAGE_KEYS},
                               "ms_device_over_mean": round(slow["ms_device"] / ideal, 4) if ideal > 0 else None}
        out["reduce_ms"] = max(r["reduce_ms"] for r in per_rank)
    status = 0

    # ---- CPU legs: baseline timing, parity frame, reference-order counters --------------------------
    # (rank 0 only; at N > 1 the other ranks wait in the parity render's reduce. `cpu_baseline` is an N = 1 entry; at
    # N > 1 the leg still runs, for the frame the merged GPU frame is compared with)
    ref_order = None
    if not args.no_cpu_baseline:
        cb, ref_frame = (None, None)
        if rank == 0:
            cb, ref_frame = cpu_baseline(scene, p, args)
        have = [cb is not None]
        if world > 1:
            dist.broadcast_object_list(have, src=0)
        if have[0]:
            pp = dict(p, spp=args.cpu_spp)
            if world == 1:
                img, _ = dscene.render(pp, flags=args.flags)
            else:
                fb2 = torch.zeros_like(fb)
                dscene.render_into(fb2, pp, rank=rank, world_size=world, flags=args.flags, stream=stream)
                merge_frame(fb2)
                torch.cuda.synchronize()
                img = fb2.cpu().numpy()
            if rank == 0:
                entry, ref_order = cb
                if world == 1:
                    out["cpu_baseline"] = entry
                e = float(np.sqrt(np.mean((np.nan_to_num(img[..., :3]).astype(np.float64) - np.nan_to_num(ref_frame[..., :3])) ** 2)))
                same = float(np.mean(np.all(img.view(np.uint32) == ref_frame.view(np.uint32), axis=-1)))
                out["parity"] = {"rmse": e, "identical_pixel_fraction": round(same, 6), "tolerance": RMSE_TOL, "ok": bool(e < RMSE_TOL),
                                 "against": entry["kind"], "sample": f"{W}x{H}, {args.cpu_spp} spp, depth {p['depth']}: the frame of the CPU leg"
                                 + ("" if world == 1 else f", GPU frame merged from {world} ranks")}
                if not e < RMSE_TOL:
                    status = 3

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    # ---- rooflines of the three large kernels (rank 0's share of the frame at N > 1) -----------------
    if not args.no_roofline and not (args.flags & 5):
        # exact test counters from the instrumented twin library (one untimed pass; the workload is
        # deterministic so the counts are those of every timed pass)
        dscene.close()           # hand its batch buffers back: the counting pass is one batch too, launch for launch
        ds2 = api.DeviceScene(scene, device=local_rank, instrumented=True)
        _, c = ds2.render(p, rank=rank, world_size=world, flags=args.flags)
        ds2.close()
        # SURVEY §8(d) per-unit figures. B_traversal = 32 N_box + 52 N_tri + 48 per ray; B_shade = 116 per hit + 64 per
        # entry + 4 C per texture tap. The traversal figures are taken twice: from the kernel's own tallies (they include
        # one candidate-mask box test per scene node per ray and the partial walks of rays handed to the general kernels)
        # and from the oracle's counters in the REFERENCE's traversal order (per-ray means of a bounded sample x the rays
        # the kernel finished); `achieved` uses the reference-order figure when it is available.
        def per_launch(k):
            ms, n = kern[k]
            return (ms / max(1, n)) * 1e-3, max(1, n // max(1, args.steps))

        entries = []
        sh_bytes = 116 * c["shaded_hits"] + 64 * c["shade_entries"] + c["texture_tap_bytes"]
        t_l, n_l = per_launch("k_wf_shade")
        entries.append({"kernel": "k_wf_shade", "bound": "hbm", "peak": HBM_PEAK_GBPS, "avg_launch_ms": round(t_l * 1e3, 3),
                        "launches_per_step": n_l, "algorithmic_bytes_per_launch": int(sh_bytes / n_l),
                        "model": "116 B x shaded hits + 64 B x entries + 4 taps x channels x texel bytes per lookup (SURVEY §8(d) B_shade)",
                        "units_per_step": {"shaded_hits": c["shaded_hits"], "entries": c["shade_entries"], "texture_tap_bytes": c["texture_tap_bytes"]}})
        for k, trav, box, tri, retried, ro in (
                ("k_wf_extend_lean", c["lean_traversals"], c["lean_box_tests"], c["lean_tri_tests"], c["retry_extend_traversals"], "closest"),
                ("k_wf_shadow_lean", c["shadow_lean_traversals"], c["shadow_lean_box_tests"], c["shadow_lean_tri_tests"], c["retry_shadow_traversals"], "shadow")):
            t_l, n_l = per_launch(k)
            own = 32 * box + 52 * tri + 48 * trav
            e = {"kernel": k, "bound": "l2", "peak": L2_PEAK_GBPS, "avg_launch_ms": round(t_l * 1e3, 3), "launches_per_step": n_l,
                 "model": "(32 B x box tests + 52 B x triangle tests + 48 B) per ray (SURVEY §8(d) B_traversal); the BVH is cache "
                          "resident, so the roof is the L2's bandwidth, not HBM's",
                 "useful_lane_ops_per_launch": int((BOX_TEST_LANE_OPS * box + TRI_TEST_LANE_OPS * tri) / n_l),
                 "kernel_tally_bytes_per_launch": int(own / n_l),
                 "kernel_tally_per_ray": {"box": round(box / max(1, trav), 2), "tri": round(tri / max(1, trav), 2)},
                 "rays_per_step": trav, "rays_handed_to_general_kernel": retried}
            alg = own
            if ref_order:
                if ro == "closest":
                    rt = ref_order["traversals"] - ref_order["shadow_traversals"]
                    rb, rtri = ref_order["box_tests"] - ref_order["shadow_box_tests"], ref_order["tri_tests"] - ref_order["shadow_tri_tests"]
                else:
                    rt, rb, rtri = ref_order["shadow_traversals"], ref_order["shadow_box_tests"], ref_order["shadow_tri_tests"]
                per_ray = 32.0 * rb / max(1, rt) + 52.0 * rtri / max(1, rt) + 48.0
                alg = per_ray * (trav - retried)
                e["reference_order_per_ray"] = {"box": round(rb / max(1, rt), 2), "tri": round(rtri / max(1, rt), 2), "bytes": round(per_ray, 1),
                                                "sample": f"oracle counters, {args.ref_order_spp} spp of the same frame"}
            e["algorithmic_bytes_per_launch"] = int(alg / n_l)
            entries.append(e)
        for e in entries:
            t = e["avg_launch_ms"] * 1e-3
            e["achieved"] = round(e["algorithmic_bytes_per_launch"] / t * 1e-9, 2) if t > 0 else 0.0
            e["unit"] = "GB/s"
            e["frac"] = round(e["achieved"] / e["peak"], 5)
            tr = pmc.get(e["kernel"]) if pmc else None
            e["traffic"] = int((tr["read_bytes"] + tr["write_bytes"]) / max(1, tr["launches"])) if tr else None
            if tr and t > 0:
                e["traffic_GBps"] = round(e["traffic"] / t * 1e-9, 1)
                e["traffic_frac_of_hbm_peak"] = round(e["traffic"] / t * 1e-9 / HBM_PEAK_GBPS, 4)
            ef = eff.get(e["kernel"]) if eff else None
            if ef:
                # how full the issued vector instructions are, and how much of the time the SIMDs issue them
                e["lane_util"] = ef["lane_util"]
                e["valu_active_per_wave_cycle"] = ef["valu_active_per_wave_cycle"]
                w = KERNEL_WAVES_PER_SIMD.get(e["kernel"])
                if w:
                    e["valu_busy"] = round(min(1.0, ef["valu_active_per_wave_cycle"] * w), 4)
                e["valu_wave_instructions_per_launch"] = ef["valu_wave_instructions"] // max(1, ef["launches"])
            if "useful_lane_ops_per_launch" in e and t > 0:
                # the VALU-side fraction: lane-operations of the box and triangle tests alone against the peak lane-operation rate
                e["valu_frac"] = round(e["useful_lane_ops_per_launch"] / t / LANE_OPS_PEAK, 4)
            if world > 1:
                e["share"] = f"rank 0 of {world}"
        dominant = max(entries, key=lambda e: e["avg_launch_ms"] * e["launches_per_step"])
        out["roofline"] = dominant
        out["rooflines"] = entries
        out["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this workload, this build, this box "
                                 "(FETCH_SIZE x 2 per MI355X_MICROARCH.md)") if pmc else None
        out["efficiency_source"] = ("rocprofv3 --pmc " + " ".join(SQ_COUNTERS) + ": one more child run of this workload; lane_util = "
                                    "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), valu_busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x "
                                    "waves per SIMD, valu_frac = (18 x box tests + 45 x triangle tests) lane-ops / launch time / "
                                    "39.3 T lane-ops/s") if eff else None
        out["stage_ms_per_step"] = {k: round(last[k], 2) for k in
                                    ("ms_extend", "ms_extend_lean", "ms_connect", "ms_shadow_lean", "ms_shade", "ms_shade_kernel", "ms_gmon", "ms_device")}
        out["counts_per_step"] = {k: c[k] for k in ("traversals", "box_tests", "tri_tests", "shaded_hits")}
    # ---- named extra + the other BASELINE configurations (N = 1 only; every handle above is closed or idle by now) ----
    if world == 1 and not args.no_secondary and not (args.flags & 5):
        try:
            dscene.close()
        except Exception:
            pass
        if not p.get("max_batch_paths", 0):
            # the headline's frame as ONE batch (rounds 1-4's configuration), same scene, same steps up to 5
            ds1 = api.DeviceScene(scene, device=local_rank)
            p1 = dict(p, max_batch_paths=min(W * H * p["spp"], (1 << 31) - 64))
            n1 = max(1, min(args.steps, 5))
            ds1.render_into(fb, p1, flags=args.flags, stream=stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n1):
                ds1.render_into(fb, p1, flags=args.flags, stream=stream)
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            ds1.close()
            out["one_batch"] = {"value": round(W * H * p["spp"] * n1 / d1 * 1e-6, 3), "unit": "Msamples/s", "steps": n1, "warmup": 1,
                                "ms_per_step": round(d1 / n1 * 1e3, 3), "max_batch_paths": int(p1["max_batch_paths"]),
                                "note": "the same frame as one batch (152 GB of path state): not what a caller gets by default"}
        del fb
        torch.cuda.empty_cache()
        sec, sec_ok = secondary_workloads(args, api, torch, np, scene, p, local_rank)
        out["secondary"] = sec
        if not sec_ok:
            status = 3
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main() or 0)
