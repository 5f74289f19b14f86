"""The RCCL branch of the multi-device merge (csrc/multi_device.inc; stands for the reference's finishTile merge,
cpu/tile-renderer.hpp:225-241) and its FAILURE paths on a one-GPU box.

libyart_hip.so binds RCCL through a table of function pointers loaded by name; YART_RCCL_LIB names the library. tests/fake_rccl is
a stand-in that implements the eight bound entry points with device-to-device copies on one GPU, logs every call and can be told to
fail the k-th call of a function. With it (and YART_MULTI_ASSUME_DISTINCT=1) the device list [0, 0, 0] takes the `distinct` branch:
ncclCommInitAll, one group of ncclSend / ncclRecv per merge, unpack — the code an 8-GPU node runs, which no round had executed with
more than one rank. Each case runs in its own process (the library is bound once per process) under a timeout: a hang is a failure.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT

FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
FAKE = os.path.join(FAKE_DIR, "_build", "libfake_rccl.so")
YART_E_RCCL = -5


@pytest.fixture(scope="module")
def fake_rccl():
    src = os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(FAKE), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-std=c++17", "-o", FAKE, src], check=True)
    return FAKE


def test_stand_in_exports_what_the_product_binds(fake_rccl):
    """The stand-in defines every symbol csrc/multi_device.inc looks up (a missing one would surface as 'librccl.so lacks ...')."""
    out = subprocess.run(["nm", "-D", "--defined-only", fake_rccl], check=True, capture_output=True, text=True).stdout
    src = open(os.path.join(ROOT, "yart_amd", "csrc", "multi_device.inc")).read()
    import re
    bound = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert len(bound) == 8
    for name in bound:
        assert f" T {name}" in out, name


CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {root!r})
from yart_amd import api
from tests.paramfile import load_params
base = {base!r}
p = dict(load_params(base + ".txt"), shard_tile=16)      # 16 blocks of the 64x64 frame: every rank owns pixels
res = {{"steps": []}}
single = api.DeviceScene(base + ".yscn", device=0)
ref, st0 = single.render(p)
single.close()
try:
    m = api.MultiDeviceScene(base + ".yscn", {devices!r})
except api.YartError as e:
    res["create_error"] = [e.code, str(e)]
    print(json.dumps(res)); sys.exit(0)
for k in range({renders}):
    try:
        img, st = m.render(p)
        res["steps"].append({{"ok": True, "identical": bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32))),
                             "rays": int(st["rays"]), "rays_single": int(st0["rays"]), "samples": int(st["samples"]),
                             "samples_single": int(st0["samples"]), "failed": m.failed_replicas()}})
    except api.YartError as e:
        res["steps"].append({{"ok": False, "code": e.code, "message": str(e)}})
if {tiles}:
    seen = []
    img, st, _ = m.render_tiles(p, on_tile=lambda frame, t: seen.append((t["x"], t["y"], t["width"], t["height"], t["rays"], t["wave"])) and None)
    res["tiles"] = {{"identical": bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32))), "n": len(seen),
                    "rays": sum(t[4] for t in seen), "rays_single": int(st0["rays"]), "distinct": len(set(t[:4] + (t[5],) for t in seen))}}
m.close()
print(json.dumps(res))
"""


def _run(fake, tmp_path, devices=(0, 0, 0), renders=1, fail=None, assume=True, case="cornell_waves", fault=None, tiles=False):
    log = str(tmp_path / "rccl.log")
    env = dict(os.environ, YART_RCCL_LIB=fake, FAKE_RCCL_LOG=log)
    if fault is not None:
        env["YART_FAULT_REPLICA"], env["YART_FAULT_AT"] = str(fault[0]), str(fault[1])
    if assume:
        env["YART_MULTI_ASSUME_DISTINCT"] = "1"
    if fail:
        env["FAKE_RCCL_FAIL_FN"], env["FAKE_RCCL_FAIL_CALL"] = fail[0], str(fail[1])
    code = CHILD.format(root=ROOT, base=os.path.join(GOLDEN, case), devices=list(devices), renders=renders, tiles=bool(tiles))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)   # (a hang fails here)
    assert r.returncode == 0, r.stderr[-2000:]
    calls = open(log).read().splitlines() if os.path.exists(log) else []
    return json.loads(r.stdout.strip().splitlines()[-1]), calls


@pytest.mark.gpu
def test_distinct_branch_gives_the_single_device_frame(fake_rccl, tmp_path):
    """Three ranks on device 0 through the RCCL branch: communicators from ncclCommInitAll, per merge ONE group holding a send and
    a receive per remote rank, and the frame of every progressive wave is the single-device frame bit for bit."""
    res, calls = _run(fake_rccl, tmp_path, renders=2)
    assert len(res["steps"]) == 2
    for s in res["steps"]:
        assert s["ok"] and s["identical"] and s["rays"] == s["rays_single"], s
    assert calls[0].startswith("ncclCommInitAll n=3")
    ends = [c for c in calls if c.startswith("ncclGroupEnd")]
    assert len(ends) == 2 and all("matched_pairs=2 unmatched=0" in c and "rc=0" in c for c in ends), ends
    assert sum(c.startswith("ncclSend") for c in calls) == 4 and sum(c.startswith("ncclRecv") for c in calls) == 4
    assert sum(c.startswith("ncclCommDestroy") for c in calls) == 3 and not any(c.startswith("ncclCommAbort") for c in calls)


@pytest.mark.gpu
@pytest.mark.parametrize("fn,call", [("send", 1), ("send", 2), ("recv", 1), ("recv", 2), ("groupend", 1)])
def test_failure_inside_the_group(fake_rccl, tmp_path, fn, call):
    """A refused call inside the merge's group -> YART_E_RCCL now, the handle marked broken (the next render is refused with
    YART_E_RCCL as well, without touching RCCL again), the communicators aborted, the thread's group CLOSED — and no hang.
    Refused send: the group holds only complete pairs, it is closed first and the communicators are aborted outside any group.
    Refused receive: the group holds a send without its receive; the communicators are aborted first, so that no unmatched
    operation is ever submitted, then the group is closed."""
    res, calls = _run(fake_rccl, tmp_path, renders=2, fail=(fn, call))
    first, second = res["steps"]
    assert not first["ok"] and first["code"] == YART_E_RCCL, first
    assert not second["ok"] and second["code"] == YART_E_RCCL and "aborted" in second["message"], second
    starts = [i for i, c in enumerate(calls) if c.startswith("ncclGroupStart")]
    ends = [i for i, c in enumerate(calls) if c.startswith("ncclGroupEnd")]
    aborts = [i for i, c in enumerate(calls) if c.startswith("ncclCommAbort")]
    assert len(starts) == 1 and len(ends) == 1 and len(aborts) == 3, calls       # one group, closed; every communicator aborted; nothing after
    assert not any(c.startswith("ncclCommDestroy") for c in calls)
    assert "unmatched=0" in calls[ends[0]], calls[ends[0]]                          # an unmatched operation is never submitted
    if fn == "recv":
        assert max(aborts) < ends[0]
        assert "dropped_on_aborted_comms=" in calls[ends[0]] and "matched_pairs=0" in calls[ends[0]]
    else:
        assert ends[0] < min(aborts)
        assert all("group_depth=0" in calls[i] and "queued_ops_on_comm=0" in calls[i] for i in aborts)
        if fn == "send":
            assert f"matched_pairs={call - 1}" in calls[ends[0]]


@pytest.mark.gpu
def test_failure_at_communicator_setup(fake_rccl, tmp_path):
    res, calls = _run(fake_rccl, tmp_path, fail=("init", 1))
    assert res.get("create_error", [0])[0] == YART_E_RCCL, res
    assert not any(c.startswith("ncclGroupStart") for c in calls)


@pytest.mark.gpu
def test_repeated_devices_without_the_override_use_peer_copies(fake_rccl, tmp_path):
    """Without YART_MULTI_ASSUME_DISTINCT a repeated device list keeps the rehearsal transport (device-to-device copies): the
    named library is never called."""
    res, calls = _run(fake_rccl, tmp_path, assume=False)
    assert res["steps"][0]["ok"] and res["steps"][0]["identical"]
    assert calls == []


@pytest.mark.gpu
def test_named_library_that_cannot_be_loaded_is_an_error(tmp_path):
    res, _ = _run(str(tmp_path / "no_such_librccl.so"), tmp_path)
    assert res.get("create_error", [0])[0] == YART_E_RCCL and "YART_RCCL_LIB" in res["create_error"][1], res


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["rccl stand-in", "peer copies"])
@pytest.mark.parametrize("replica,at", [(1, 0), (2, 1)])
def test_failed_device_is_taken_out_of_service_and_its_blocks_requeued(fake_rccl, tmp_path, transport, replica, at):
    """SURVEY §5 "failure detection": per-GPU failure = re-queue that GPU's tiles (they are idempotent). A HIP error on the thread of
    replica 1 or 2 (injected: YART_FAULT_REPLICA / YART_FAULT_AT, at the first or the second render of the handle) does not fail the
    call: the replica is out of service from then on, its pixel blocks are rendered on devices[0] after the merge — through the
    RCCL branch (the failed rank posts no send and gets no receive) and through the one-device transport alike — and every frame,
    ray count and sample count stays the single device's; the tile callbacks still report every block of the frame once per wave
    with its own ray count."""
    res, calls = _run(fake_rccl, tmp_path, renders=3, assume=(transport == "rccl stand-in"), fault=(replica, at), tiles=True)
    assert len(res["steps"]) == 3
    for k, s in enumerate(res["steps"]):
        assert s["ok"] and s["identical"] and s["rays"] == s["rays_single"] and s["samples"] == s["samples_single"], (k, s)
        assert s["failed"] == ([replica] if k >= at else []), (k, s)
    t = res["tiles"]
    assert t["identical"] and t["rays"] == t["rays_single"] and t["n"] == t["distinct"] == 16 * 2, t     # 16 blocks x 2 waves
    if transport == "rccl stand-in":
        ends = [c for c in calls if c.startswith("ncclGroupEnd")]
        assert all("unmatched=0" in c and "rc=0" in c for c in ends), ends
        assert [("matched_pairs=2" in c) for c in ends][:at] == [True] * at              # both remote ranks before the failure,
        assert all("matched_pairs=1" in c for c in ends[at:]), ends                       # one afterwards
        assert not any(c.startswith("ncclCommAbort") for c in calls)


@pytest.mark.gpu
def test_failure_of_the_merge_device_is_the_calls_error(fake_rccl, tmp_path):
    res, _ = _run(fake_rccl, tmp_path, renders=1, fault=(0, 0))
    assert not res["steps"][0]["ok"] and res["steps"][0]["code"] == -3, res            # YART_E_HIP
