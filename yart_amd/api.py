"""ctypes binding of ``libyart_hip.so`` and a host-side mirror of the reference's
renderer interface.

The reference's seam is the abstract class ``yart::Renderer`` (reference
``src/core/renderer.hpp:17-104``) implemented by ``yart::cpu::TileRenderer``
(``src/cpu/tile-renderer.hpp:22-310``).  :class:`HipTileRenderer` keeps that
surface — public knobs ``samples / first_wave_samples / max_wave_samples /
tile_size / background_color / scene``, methods ``render() / abort() / wait() /
render_sync()`` returning a ``RenderData`` — on top of the C ABI declared in
``include/yart_hip.h``.

There is no CPU fallback: if the shared library is missing, or no HIP device is
visible, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import yscn

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyart_hip.so")
if os.environ.get("YART_LIB_VARIANT"):      # experiment builds of tools/build_variant.sh (yart_amd/_variants/NAME.so)
    LIB_PATH = os.path.join(_HERE, "_variants", os.environ["YART_LIB_VARIANT"] + ".so")

YART_OK, YART_E_INVALID, YART_E_NO_DEVICE, YART_E_HIP, YART_E_IO, YART_E_RCCL = 0, -1, -2, -3, -4, -5
ABI_VERSION = 3          # include/yart_hip.h: YART_HIP_ABI_VERSION (struct layouts below)
FLAG_MEGAKERNEL = 1
FLAG_SHADE_SORT = 2
FLAG_GENERAL_TRACE = 4
FLAG_DIRECT_SAMPLER = 8
FLAG_NO_REFILL = 16
FLAG_NO_COMPACTION = 32
FLAG_NO_SHADE_SORT = 64
FLAG_NO_RESUME = 128
FLAG_PATH_POOL = 512


class YartError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"yart_hip error {code}: {message}")
        self.code = code


# ---------------------------------------------------------------------------
# POD mirrors of include/yart_hip.h
# ---------------------------------------------------------------------------
class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32),
                ("is_float", C.c_uint32), ("type", C.c_uint32), ("data", C.c_void_p)]


class MaterialDesc(C.Structure):
    _fields_ = [("base", C.c_float * 3), ("emission", C.c_float * 3),
                ("metallic", C.c_float), ("roughness", C.c_float), ("transmission", C.c_float),
                ("ior", C.c_float), ("anisotropic", C.c_float), ("aniso_rotation", C.c_float),
                ("clearcoat", C.c_float), ("clearcoat_roughness", C.c_float),
                ("normal_scale", C.c_float), ("thin_transmission", C.c_uint32),
                ("volume_color", C.c_float * 3), ("volume_density", C.c_float),
                ("tex_base", C.c_int32), ("tex_mr", C.c_int32), ("tex_transmission", C.c_int32),
                ("tex_normal", C.c_int32), ("tex_clearcoat", C.c_int32), ("tex_emission", C.c_int32)]


class MeshDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_faces", C.c_uint32),
                ("positions", C.c_void_p), ("normals", C.c_void_p), ("tangents", C.c_void_p),
                ("uvs", C.c_void_p), ("faces", C.c_void_p), ("face_light", C.c_void_p)]


class NodeDesc(C.Structure):
    _fields_ = [("parent", C.c_int32), ("mesh", C.c_int32), ("fwd", C.c_float * 16), ("inv", C.c_float * 16)]


class LightDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("mesh", C.c_int32), ("tri", C.c_uint32), ("two_sided", C.c_uint32),
                ("texture", C.c_int32), ("radius", C.c_float), ("emission", C.c_float * 3),
                ("fwd", C.c_float * 16), ("inv", C.c_float * 16)]


class SceneDesc(C.Structure):
    _fields_ = [("n_textures", C.c_uint32), ("n_materials", C.c_uint32), ("n_meshes", C.c_uint32),
                ("n_nodes", C.c_uint32), ("n_lights", C.c_uint32),
                ("textures", C.POINTER(TextureDesc)), ("materials", C.POINTER(MaterialDesc)),
                ("meshes", C.POINTER(MeshDesc)), ("nodes", C.POINTER(NodeDesc)),
                ("lights", C.POINTER(LightDesc))]


class CameraDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("focal_length", C.c_float),
                ("f_number", C.c_float), ("sensor", C.c_float * 2), ("position", C.c_float * 3),
                ("target", C.c_float * 3), ("up", C.c_float * 3), ("exposure", C.c_float),
                ("aperture_sides", C.c_uint32)]


class RenderParams(C.Structure):
    _fields_ = [("samples", C.c_uint32), ("first_wave_samples", C.c_uint32),
                ("max_wave_samples", C.c_uint32), ("tile_size", C.c_uint32), ("max_depth", C.c_uint32),
                ("background", C.c_float * 3), ("rank", C.c_uint32), ("world_size", C.c_uint32),
                ("flags", C.c_uint32), ("start_sample", C.c_uint32), ("stop_sample", C.c_uint32),
                ("estimator", C.c_uint32), ("shard_tile", C.c_uint32), ("max_batch_paths", C.c_uint32),
                ("pool_paths", C.c_uint32)]


def _plain(v):
    return list(v) if hasattr(v, "__len__") else v


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("ms_total", C.c_double),
                ("ms_device", C.c_double), ("ms_traverse", C.c_double), ("traversals", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("waves", C.c_uint32),
                ("launches_traverse", C.c_uint32), ("shaded_hits", C.c_uint64),
                ("ms_extend", C.c_double), ("ms_shade", C.c_double), ("ms_connect", C.c_double),
                ("ms_gmon", C.c_double), ("launches_extend", C.c_uint32), ("launches_connect", C.c_uint32),
                ("ms_extend_lean", C.c_double), ("lean_traversals", C.c_uint64), ("lean_box_tests", C.c_uint64),
                ("lean_tri_tests", C.c_uint64), ("launches_extend_lean", C.c_uint32), ("reserved0", C.c_uint32),
                ("ms_shade_kernel", C.c_double), ("ms_shadow_lean", C.c_double),
                ("launches_shade_kernel", C.c_uint32), ("launches_shadow_lean", C.c_uint32),
                ("shadow_lean_traversals", C.c_uint64), ("shadow_lean_box_tests", C.c_uint64),
                ("shadow_lean_tri_tests", C.c_uint64), ("shade_entries", C.c_uint64),
                ("texture_tap_bytes", C.c_uint64), ("pipeline_flags", C.c_uint32), ("reserved1", C.c_uint32),
                ("retry_extend_traversals", C.c_uint64), ("retry_shadow_traversals", C.c_uint64),
                ("wide_extend_nodes", C.c_uint64), ("wide_extend_tris", C.c_uint64),
                ("wide_shadow_nodes", C.c_uint64), ("wide_shadow_tris", C.c_uint64),
                ("wide_extend_handed", C.c_uint64 * 4), ("wide_shadow_handed", C.c_uint64 * 4),
                ("paths_at_bounce", C.c_uint64 * 16)]

    def asdict(self):
        return {k: _plain(getattr(self, k)) for k, _ in self._fields_}


# YartRenderParams.estimator (core/estimator.hpp; the reference picks one at compile time, integrator.cpp:17-18)
ESTIMATOR_GMON, ESTIMATOR_MEAN, ESTIMATOR_MON, ESTIMATOR_GMONB = 0, 1, 2, 3


YART_ABORTED = 1
# int on_wave(void* user, const YartStats* wave_stats, wave, wave_samples, samples_taken, total_samples)
WAVE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)


class TileInfo(C.Structure):
    """YartTileInfo (include/yart_hip.h): Renderer::TileData of a finished pixel block (renderer.hpp:40-50)."""
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("index", C.c_uint32), ("total", C.c_uint32), ("wave", C.c_uint32), ("wave_samples", C.c_uint32),
                ("samples_taken", C.c_uint32), ("total_samples", C.c_uint32), ("rays", C.c_uint64), ("ms", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# int on_tile(void* user, const YartTileInfo* tile)
TILE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(TileInfo))


class ImportOptions(C.Structure):
    """YartImportOptions (include/yart_hip.h): the environment the frontend adds after gltf::load."""
    _fields_ = [("env_hdr_path", C.c_char_p), ("env_radius", C.c_float), ("uniform_env", C.c_uint32),
                ("uniform_emission", C.c_float * 3), ("reserved", C.c_uint32 * 4)]


def import_options(env_hdr=None, env_radius=100.0, uniform_env=None) -> ImportOptions:
    o = ImportOptions()
    o.env_hdr_path = os.fspath(env_hdr).encode() if env_hdr else None
    o.env_radius = float(env_radius)
    if uniform_env is not None:
        o.uniform_env = 1
        o.uniform_emission = _f(uniform_env, 3)
    return o


def is_gltf_path(path) -> bool:
    return os.fspath(path).lower().endswith((".glb", ".gltf"))


def gltf_to_yscn(gltf_path, yscn_path, env_hdr=None, env_radius=100.0, uniform_env=None):
    """Import a glTF 2.0 / GLB asset the way the reference's loader does (src/gltf/gltf.cpp:319-358, plus the
    environment of src/main.cpp:80-86) and write it as a ``.yscn`` container. Host only: needs no device."""
    L = lib()
    o = import_options(env_hdr, env_radius, uniform_env)
    _check(L.yart_hip_gltf_to_yscn(os.fspath(gltf_path).encode(), C.byref(o), os.fspath(yscn_path).encode()), L)


EXPORTS = ["yart_hip_abi_version", "yart_hip_device_count", "yart_hip_last_error",
           "yart_hip_scene_create", "yart_hip_scene_load", "yart_hip_scene_destroy",
           "yart_hip_scene_load_gltf", "yart_hip_gltf_to_yscn",
           "yart_hip_render", "yart_hip_render_waves", "yart_hip_render_tiles", "yart_hip_render_device", "yart_hip_probe_samples",
           "yart_hip_probe_hits", "yart_hip_probe_sampler", "yart_hip_bvh_info", "yart_hip_bvh_copy", "yart_hip_scene_create_flags", "yart_hip_bvh_build_device", "yart_hip_bvh_build_host", "yart_hip_debug_counters", "yart_hip_debug_shade_regions",
           "yart_hip_tonemap_agx", "yart_hip_encode_rgb8", "yart_hip_tonemap_host",
           "yart_hip_multi_create", "yart_hip_multi_load", "yart_hip_multi_destroy", "yart_hip_multi_device_count", "yart_hip_multi_failed_devices",
           "yart_hip_multi_render", "yart_hip_multi_render_tiles", "yart_hip_multi_rccl_selftest"]

LIB_COUNT_PATH = os.path.join(_HERE, "libyart_hip_count.so")   # instrumented twin (exact test counters)
_libs = {}


def lib(instrumented: bool = False):
    """Load libyart_hip.so (built by ``__graft_entry__.build()``); raises if absent.
    ``instrumented=True`` loads libyart_hip_count.so, the same code compiled with
    -DYART_COUNT_TRAVERSAL (exact box / triangle test counters; never timed)."""
    path = LIB_COUNT_PATH if instrumented else LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise YartError(YART_E_NO_DEVICE, f"{path} not built (run __graft_entry__.build()); "
                                              "there is no CPU fallback")
        L = C.CDLL(path)
        if L.yart_hip_abi_version() != ABI_VERSION:
            raise YartError(YART_E_INVALID, f"{path} has ABI {L.yart_hip_abi_version()}, this module binds ABI {ABI_VERSION} "
                                            "(include/yart_hip.h: YART_HIP_ABI_VERSION); rebuild the library")
        L.yart_hip_last_error.restype = C.c_char_p
        L.yart_hip_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.yart_hip_scene_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.yart_hip_scene_load_gltf.argtypes = [C.c_char_p, C.POINTER(ImportOptions), C.c_int, C.POINTER(C.c_void_p)]
        L.yart_hip_gltf_to_yscn.argtypes = [C.c_char_p, C.POINTER(ImportOptions), C.c_char_p]
        L.yart_hip_scene_destroy.argtypes = [C.c_void_p]
        L.yart_hip_scene_destroy.restype = None
        L.yart_hip_render.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams),
                                      C.c_void_p, C.POINTER(Stats)]
        L.yart_hip_render_waves.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p,
                                            C.POINTER(Stats), WAVE_CALLBACK, C.c_void_p]
        L.yart_hip_render_tiles.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p,
                                            C.POINTER(Stats), WAVE_CALLBACK, TILE_CALLBACK, C.c_void_p]
        L.yart_hip_render_device.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams),
                                             C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.yart_hip_probe_samples.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams),
                                             C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.yart_hip_probe_hits.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.yart_hip_probe_sampler.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.yart_hip_tonemap_agx.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.yart_hip_encode_rgb8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.yart_hip_tonemap_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.yart_hip_multi_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]
        L.yart_hip_multi_load.argtypes = [C.c_char_p, C.POINTER(ImportOptions), C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]
        L.yart_hip_multi_destroy.argtypes = [C.c_void_p]
        L.yart_hip_multi_destroy.restype = None
        L.yart_hip_multi_device_count.argtypes = [C.c_void_p]
        L.yart_hip_multi_render.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
        L.yart_hip_multi_render_tiles.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p,
                                                  C.POINTER(Stats), WAVE_CALLBACK, TILE_CALLBACK, C.c_void_p]
        L.yart_hip_bvh_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.yart_hip_bvh_copy.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _libs[path] = L
    return _libs[path]


def _check(code, L=None):
    if code != YART_OK:
        raise YartError(code, (L or lib()).yart_hip_last_error().decode())


def bvh_build(positions, faces, device=None, threads=0):
    """The reference's binned-SAH BVH of one mesh, built outside a scene: on HIP device ``device`` (csrc/bvh_build_device.inc)
    or, with ``device=None``, on the host (csrc/bvh_build.hpp, ``threads`` workers, 0 = all). Returns (nodes (n, 8) u32 —
    bounds as float bits, left|first, span —, indices (n_faces,) u32, milliseconds); both builds give the same bytes."""
    L = lib()
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    f = np.ascontiguousarray(faces, np.uint32)
    f = f.reshape(len(f), -1)
    nodes = np.zeros((2 * len(f), 8), np.uint32)
    idx = np.zeros(len(f), np.uint32)
    n, ms = C.c_uint32(0), C.c_double(0)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    if device is None:
        L.yart_hip_bvh_build_host.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        _check(L.yart_hip_bvh_build_host(vp(pos), len(pos), vp(f), f.shape[1], len(f), threads, vp(nodes), vp(idx), C.byref(n), C.byref(ms)), L)
    else:
        L.yart_hip_bvh_build_device.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        _check(L.yart_hip_bvh_build_device(int(device), vp(pos), len(pos), vp(f), f.shape[1], len(f), vp(nodes), vp(idx), C.byref(n), C.byref(ms)), L)
    return nodes[:n.value].copy(), idx, ms.value


def _f(arr, n):
    return (C.c_float * n)(*[float(v) for v in np.asarray(arr, np.float32).reshape(-1)[:n]])


def make_camera(p: dict) -> CameraDesc:
    """``p`` uses the vocabulary of oracle/params.hpp / yart_amd.scenes."""
    c = CameraDesc()
    c.width, c.height = int(p["size"][0]), int(p["size"][1])
    c.focal_length = float(p.get("focal", 35.0)); c.f_number = float(p.get("fnumber", 0.0))
    c.sensor = _f(p.get("sensor", (36.0, 24.0)), 2)
    c.position = _f(p["eye"], 3); c.target = _f(p["target"], 3); c.up = _f(p.get("up", (0, 1, 0)), 3)
    c.exposure = float(p.get("exposure", 0.0)); c.aperture_sides = int(p.get("aperture_sides", 0))
    return c


def make_params(p: dict, rank=0, world_size=1, flags=0) -> RenderParams:
    r = RenderParams()
    r.samples = int(p["spp"])
    r.first_wave_samples = int(p.get("first_wave", p["spp"]))     # single wave, as main.cpp:97-99
    r.max_wave_samples = int(p.get("max_wave", p["spp"]))
    r.tile_size = int(p.get("tile", 64)); r.max_depth = int(p.get("depth", 30))
    r.background = _f(p.get("background", (0, 0, 0)), 3)
    r.rank, r.world_size, r.flags = int(rank), int(world_size), int(flags)
    r.start_sample, r.stop_sample = int(p.get("start_sample", 0)), int(p.get("stop_sample", 0))
    r.estimator = int(p.get("estimator", ESTIMATOR_GMON))
    r.shard_tile = int(p.get("shard_tile", 0))
    r.max_batch_paths = int(p.get("max_batch_paths", 0))
    r.pool_paths = int(p.get("pool_paths", 0))
    return r


class DeviceScene:
    """Owns a ``YartScene*`` (device-resident flattened scene + BVHs)."""

    def __init__(self, scene, device: int = -1, instrumented: bool = False, env_hdr=None, env_radius=100.0,
                 uniform_env=None, host_bvh: bool = False):
        """scene: a :class:`yscn.Scene`, the path of a ``.yscn`` container, or the path of a ``.glb`` / ``.gltf``
        asset (then env_hdr / env_radius / uniform_env give the environment light, as main.cpp:80-86)."""
        self._h = C.c_void_p()
        self._keep = []
        self._L = lib(instrumented)
        if isinstance(scene, (str, os.PathLike)) and is_gltf_path(scene):
            o = import_options(env_hdr, env_radius, uniform_env)
            _check(self._L.yart_hip_scene_load_gltf(os.fspath(scene).encode(), C.byref(o), device, C.byref(self._h)),
                   self._L)
        elif isinstance(scene, (str, os.PathLike)):
            _check(self._L.yart_hip_scene_load(os.fspath(scene).encode(), device, C.byref(self._h)), self._L)
        else:
            desc = self._describe(scene)
            if host_bvh:      # YART_SCENE_HOST_BVH: the meshes' BVHs from the host builder instead of the device build (same bytes)
                self._L.yart_hip_scene_create_flags.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
                _check(self._L.yart_hip_scene_create_flags(C.byref(desc), device, 2, C.byref(self._h)), self._L)
            else:
                _check(self._L.yart_hip_scene_create(C.byref(desc), device, C.byref(self._h)), self._L)
        self._keep = []     # the library copies everything it needs

    def _describe(self, s: yscn.Scene) -> SceneDesc:
        keep = self._keep

        def ptr(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype); keep.append(a)
            return a.ctypes.data_as(C.c_void_p)
        tex = (TextureDesc * max(len(s.textures), 1))()
        for i, t in enumerate(s.textures):
            tex[i] = TextureDesc(t.width, t.height, t.channels, int(t.is_float), t.type,
                                 ptr(t.data, np.float32 if t.is_float else np.uint8))
        mats = (MaterialDesc * max(len(s.materials), 1))()
        for i, m in enumerate(s.materials):
            C.memmove(C.byref(mats[i]), m.pack(), C.sizeof(MaterialDesc))
        meshes = (MeshDesc * max(len(s.meshes), 1))()
        for i, m in enumerate(s.meshes):
            meshes[i] = MeshDesc(len(m.positions), len(m.faces), ptr(m.positions, np.float32),
                                 ptr(m.normals, np.float32), ptr(m.tangents, np.float32),
                                 ptr(m.uvs, np.float32), ptr(m.faces, np.uint32), ptr(m.face_light, np.int32))
        nodes = (NodeDesc * len(s.nodes))()
        for i, n in enumerate(s.nodes):
            nodes[i] = NodeDesc(n.parent, n.mesh, _f(n.fwd, 16), _f(n.inv, 16))
        lights = (LightDesc * max(len(s.lights), 1))()
        for i, l in enumerate(s.lights):
            lights[i] = LightDesc(l.type, l.mesh, l.tri, int(l.two_sided), l.texture, l.radius,
                                  _f(l.emission, 3), _f(l.fwd, 16), _f(l.inv, 16))
        keep += [tex, mats, meshes, nodes, lights]
        return SceneDesc(len(s.textures), len(s.materials), len(s.meshes), len(s.nodes), len(s.lights),
                         tex, mats, meshes, nodes, lights)

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self._L.yart_hip_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- rendering ----------------------------------------------------------------
    def render(self, p: dict, rank=0, world_size=1, flags=0, accumulated=None):
        """Blocking render to a host array (H, W, 4) float32 + stats dict. `accumulated`: the frame of the
        samples before p["start_sample"] when resuming (YartRenderParams.start_sample)."""
        cam, rp, st = make_camera(p), make_params(p, rank, world_size, flags), Stats()
        out = np.empty((cam.height, cam.width, 4), np.float32)
        if accumulated is not None:
            out[...] = accumulated
        _check(self._L.yart_hip_render(self._h, C.byref(cam), C.byref(rp), out.ctypes.data_as(C.c_void_p),
                                     C.byref(st)))
        return out, st.asdict()

    def render_waves(self, p: dict, on_wave=None, rank=0, world_size=1, flags=0, accumulated=None):
        """Like :meth:`render`, one wave of the schedule at a time (tile-renderer.hpp:264-289).
        ``on_wave(frame, info)`` is called after every wave with the frame blended so far (a view of the output
        array) and ``info = dict(wave, wave_samples, samples_taken, total_samples)``; returning a true value stops
        the render after that wave (Renderer::abort). Returns (frame, stats, aborted)."""
        cam, rp, st = make_camera(p), make_params(p, rank, world_size, flags), Stats()
        out = np.empty((cam.height, cam.width, 4), np.float32)
        if accumulated is not None:
            out[...] = accumulated
        errors = []

        def trampoline(_user, _stats, wave, wave_samples, taken, total):
            try:
                stop = on_wave(out, dict(wave=wave, wave_samples=wave_samples, samples_taken=taken, total_samples=total))
                return 1 if stop else 0
            except BaseException as e:          # an exception must not unwind through the C frames
                errors.append(e)
                return 1
        cb = WAVE_CALLBACK(trampoline) if on_wave else WAVE_CALLBACK()
        rc = self._L.yart_hip_render_waves(self._h, C.byref(cam), C.byref(rp), out.ctypes.data_as(C.c_void_p),
                                           C.byref(st), cb, None)
        if errors:
            raise errors[0]
        if rc != YART_ABORTED:
            _check(rc, self._L)
        return out, st.asdict(), rc == YART_ABORTED

    def render_tiles(self, p: dict, on_tile=None, on_wave=None, rank=0, world_size=1, flags=0, accumulated=None):
        """Like :meth:`render_waves`, with tile granularity (Renderer::onRenderTileComplete): ``on_tile(frame, tile)``
        is called for every pixel block a batch of a wave has finished, ``frame`` being the output array (holding the
        blended values of that block) and ``tile`` a dict of YartTileInfo; a true return stops the render.
        ``p["max_batch_paths"]`` bounds a batch, i.e. how many blocks arrive together. Returns (frame, stats, aborted)."""
        cam, rp, st = make_camera(p), make_params(p, rank, world_size, flags), Stats()
        out = np.empty((cam.height, cam.width, 4), np.float32)
        if accumulated is not None:
            out[...] = accumulated
        errors = []

        def wave_tr(_user, _stats, wave, wave_samples, taken, total):
            try:
                ws = C.cast(_stats, C.POINTER(Stats)).contents if _stats else None
                return 1 if on_wave(out, dict(wave=wave, wave_samples=wave_samples, samples_taken=taken, total_samples=total,
                                              rays=int(ws.rays) if ws is not None else 0)) else 0
            except BaseException as e:
                errors.append(e)
                return 1

        def tile_tr(_user, tile):
            try:
                return 1 if on_tile(out, tile.contents.asdict()) else 0
            except BaseException as e:
                errors.append(e)
                return 1
        wcb = WAVE_CALLBACK(wave_tr) if on_wave else WAVE_CALLBACK()
        tcb = TILE_CALLBACK(tile_tr) if on_tile else TILE_CALLBACK()
        rc = self._L.yart_hip_render_tiles(self._h, C.byref(cam), C.byref(rp), out.ctypes.data_as(C.c_void_p),
                                           C.byref(st), wcb, tcb, None)
        if errors:
            raise errors[0]
        if rc != YART_ABORTED:
            _check(rc, self._L)
        return out, st.asdict(), rc == YART_ABORTED

    def render_into(self, tensor, p: dict, rank=0, world_size=1, flags=0, stream=None):
        """Render into a CUDA/HIP torch tensor of shape (H, W, 4) float32 (device memory)."""
        cam, rp, st = make_camera(p), make_params(p, rank, world_size, flags), Stats()
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == cam.width * cam.height * 4
        sp = C.c_void_p(stream) if stream else None
        _check(self._L.yart_hip_render_device(self._h, C.byref(cam), C.byref(rp),
                                            C.c_void_p(tensor.data_ptr()), sp, C.byref(st)))
        return st.asdict()

    # -- diagnostics ----------------------------------------------------------------
    def probe_samples(self, p: dict, xys: Sequence[Sequence[int]]):
        cam, rp = make_camera(p), make_params(p)
        a = np.ascontiguousarray(xys, np.uint32).reshape(-1, 3)
        out = np.empty((len(a), 3), np.float32)
        rays = C.c_uint64()
        _check(self._L.yart_hip_probe_samples(self._h, C.byref(cam), C.byref(rp), len(a),
                                            a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                            C.byref(rays)))
        return out, rays.value

    def probe_hits(self, rays):
        a = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.empty((len(a), 16), np.float32)
        _check(self._L.yart_hip_probe_hits(self._h, len(a), a.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p)))
        return out

    def probe_sampler(self, spp, tile, cases, pattern, use_tables=False):
        """Device-side sampler draws (diagnostic): cases = [(px, py, sample)], pattern = sequence of 1 (get1D) / 2 (get2D);
        returns float32 [n_cases, sum(pattern)]."""
        cs = np.ascontiguousarray(np.asarray(cases, np.uint32).reshape(-1, 3))
        pat = np.ascontiguousarray(np.asarray(pattern, np.uint8))
        out = np.empty((len(cs), int(pat.sum())), np.float32)
        _check(self._L.yart_hip_probe_sampler(self._h, int(spp), int(tile), len(cs), cs.ctypes.data_as(C.c_void_p), len(pat),
                                              pat.ctypes.data_as(C.c_void_p), 1 if use_tables else 0, out.ctypes.data_as(C.c_void_p)), self._L)
        return out

    def debug_counters(self):
        out = (C.c_uint64 * 32)()
        self._L.yart_hip_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
        _check(self._L.yart_hip_debug_counters(self._h, out), self._L)
        return [int(v) for v in out]

    def bvh(self, mesh: int):
        nn, nt = C.c_uint32(), C.c_uint32()
        _check(self._L.yart_hip_bvh_info(self._h, mesh, C.byref(nn), C.byref(nt)))
        nodes = np.empty((nn.value, 8), np.uint32); idx = np.empty(nt.value, np.uint32)
        _check(self._L.yart_hip_bvh_copy(self._h, mesh, nodes.ctypes.data_as(C.c_void_p),
                                       idx.ctypes.data_as(C.c_void_p)))
        return nodes, idx


class MultiDeviceScene:
    """Owns a ``YartMulti*``: the scene replicated on several GPUs of this node, one host thread per device, pixel
    blocks dealt round-robin, the devices' own pixels merged on ``devices[0]`` with RCCL send / recv (include/yart_hip.h;
    stands where TileRenderer's worker pool and finishTile's merge stand, tile-renderer.hpp:150-197, 225-241)."""

    def __init__(self, scene, devices: Sequence[int], env_hdr=None, env_radius=100.0, uniform_env=None):
        self._h = C.c_void_p()
        self._L = lib()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        if isinstance(scene, (str, os.PathLike)):
            o = import_options(env_hdr, env_radius, uniform_env)
            _check(self._L.yart_hip_multi_load(os.fspath(scene).encode(), C.byref(o), devs, len(devices), C.byref(self._h)), self._L)
        else:
            helper = DeviceScene.__new__(DeviceScene)
            helper._keep = []
            desc = DeviceScene._describe(helper, scene)
            _check(self._L.yart_hip_multi_create(C.byref(desc), devs, len(devices), C.byref(self._h)), self._L)

    @property
    def n_devices(self):
        return int(self._L.yart_hip_multi_device_count(self._h))

    def failed_replicas(self):
        """Replicas taken out of service by a device failure (their blocks are rendered on devices[0]); [] normally."""
        out = (C.c_int * 64)()
        self._L.yart_hip_multi_failed_devices.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        n = int(self._L.yart_hip_multi_failed_devices(self._h, out, 64))
        return [int(out[k]) for k in range(min(n, 64))]

    def render(self, p: dict, rank=0, world_size=1, flags=0, accumulated=None):
        cam, rp, st = make_camera(p), make_params(p, rank, world_size, flags), Stats()
        out = np.empty((cam.height, cam.width, 4), np.float32)
        if accumulated is not None:
            out[...] = accumulated
        _check(self._L.yart_hip_multi_render(self._h, C.byref(cam), C.byref(rp), out.ctypes.data_as(C.c_void_p), C.byref(st)), self._L)
        return out, st.asdict()

    def render_tiles(self, p: dict, on_tile=None, on_wave=None, rank=0, world_size=1, flags=0, accumulated=None):
        """The progressive form (yart_hip_multi_render_tiles): every wave rendered by all devices, merged, reported through
        ``on_wave(frame, info)``; with ``on_tile(frame, tile)`` every block of the frame once per wave (Morton order, its own
        ray count). Returns (frame, stats, aborted)."""
        cam, rp, st = make_camera(p), make_params(p, rank, world_size, flags), Stats()
        out = np.empty((cam.height, cam.width, 4), np.float32)
        if accumulated is not None:
            out[...] = accumulated
        errors = []

        def wave_tr(_user, _stats, wave, wave_samples, taken, total):
            try:
                ws = C.cast(_stats, C.POINTER(Stats)).contents if _stats else None
                return 1 if on_wave(out, dict(wave=wave, wave_samples=wave_samples, samples_taken=taken, total_samples=total,
                                              rays=int(ws.rays) if ws is not None else 0)) else 0
            except BaseException as e:
                errors.append(e)
                return 1

        def tile_tr(_user, tile):
            try:
                return 1 if on_tile(out, tile.contents.asdict()) else 0
            except BaseException as e:
                errors.append(e)
                return 1
        wcb = WAVE_CALLBACK(wave_tr) if on_wave else WAVE_CALLBACK()
        tcb = TILE_CALLBACK(tile_tr) if on_tile else TILE_CALLBACK()
        rc = self._L.yart_hip_multi_render_tiles(self._h, C.byref(cam), C.byref(rp), out.ctypes.data_as(C.c_void_p), C.byref(st), wcb, tcb, None)
        if errors:
            raise errors[0]
        if rc != YART_ABORTED:
            _check(rc, self._L)
        return out, st.asdict(), rc == YART_ABORTED

    def close(self):
        if self._h:
            self._L.yart_hip_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class RenderData:
    """Mirror of ``yart::Renderer::RenderData`` (reference src/core/renderer.hpp:22-28)."""
    buffer: np.ndarray
    samples_taken: int
    total_samples: int
    total_rays: int
    total_time_ms: float
    stats: dict


class HipTileRenderer:
    """Drop-in for ``yart::cpu::TileRenderer<SobolSampler<FastOwenScrambler>, MISIntegrator>``
    (reference src/cpu/tile-renderer.hpp:25-115): same knobs, same blocking / async calls.

    ``camera`` is a dict in the vocabulary of ``oracle/params.hpp`` (size, focal, fnumber,
    sensor, eye, target, up, exposure, aperture_sides)."""

    def __init__(self, width: int, height: int, camera: dict, device: int = -1):
        self.samples = 64                 # DEFAULT_SAMPLE_COUNT, tile-renderer.hpp:10-13
        self.first_wave_samples = 64
        self.max_wave_samples = 128
        self.tile_size = 64
        self.max_depth = 30               # RayIntegrator::m_maxDepth, ray-integrator.hpp:14
        self.background_color = (0.0, 0.0, 0.0)
        self.scene: Optional[DeviceScene] = None
        self.tonemapper = None            # AgX look name ("none" / "golden" / "punchy") or None: linear HDR buffer
        self.estimator = ESTIMATOR_GMON   # integrator.cpp:17-18 fixes it at compile time
        self.on_render_complete = None    # callbacks of renderer.hpp:55-58
        self.on_render_aborted = None
        self.on_render_wave_complete = None   # f(RenderData, dict(wave, wave_samples, samples_taken, total_samples))
        self.on_render_tile_complete = None   # f(RenderData, dict of YartTileInfo): per finished pixel block of a batch
        self.max_batch_paths = 0              # YartRenderParams.max_batch_paths: how many blocks finish together (0: all)
        self._camera = dict(camera, size=(width, height))
        self._device = device
        self._thread: Optional[threading.Thread] = None
        self._abort = False
        self._result: Optional[RenderData] = None

    def _params(self):
        return dict(self._camera, spp=self.samples, first_wave=min(self.first_wave_samples, self.samples),
                    max_wave=self.max_wave_samples, tile=self.tile_size, depth=self.max_depth,
                    background=self.background_color, estimator=self.estimator, max_batch_paths=self.max_batch_paths)

    def render_sync(self) -> RenderData:
        if self.scene is None:           # Integrator::render: "if (!scene) return" (integrator.cpp:6)
            w, h = self._camera["size"]
            return RenderData(np.zeros((h, w, 4), np.float32), 0, self.samples, 0, 0.0, {})
        t0 = time.perf_counter()
        taken = [0]
        self._abort = False

        def shown(frame):                 # tile-renderer.hpp:234-239: the exposed buffer is the tonemapped one
            return tonemap(frame, self.tonemapper)[0] if self.tonemapper else frame

        def on_wave(frame, info):
            taken[0] = info["samples_taken"]
            if self.on_render_wave_complete:
                self.on_render_wave_complete(RenderData(shown(frame), info["samples_taken"], self.samples, 0,
                                                        (time.perf_counter() - t0) * 1e3, {}), info)
            return self._abort
        if self.on_render_tile_complete:
            display = [None]               # ONE persistent display buffer per render (integration/hip-renderer.hpp::expose does the same)

            def on_tile(frame, tile):
                x, y, w, h = tile["x"], tile["y"], tile["width"], tile["height"]
                view = frame
                if self.tonemapper:        # tile-renderer.hpp:234-239 maps the finished tile only, into the buffer it exposes
                    if display[0] is None:
                        display[0] = np.zeros_like(frame)
                    view = display[0]
                    view[y:y + h, x:x + w] = tonemap(np.ascontiguousarray(frame[y:y + h, x:x + w]), self.tonemapper)[0]
                self.on_render_tile_complete(RenderData(view, tile["samples_taken"] - tile["wave_samples"], self.samples, 0,
                                                        (time.perf_counter() - t0) * 1e3, {}), tile)
                return self._abort
            buf, st, _ = self.scene.render_tiles(self._params(), on_tile, on_wave)
        else:
            buf, st, _ = self.scene.render_waves(self._params(), on_wave)
        ms = (time.perf_counter() - t0) * 1e3
        self._result = RenderData(shown(buf), taken[0], self.samples, st["rays"], ms, st)
        return self._result

    def render(self):
        def run():
            r = self.render_sync()
            cb = self.on_render_aborted if self._abort else self.on_render_complete
            if cb:
                cb(r)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def abort(self):
        self._abort = True               # takes effect after the wave in flight (the reference: after the tiles in flight)

    def wait(self):
        if self._thread:
            self._thread.join()


AGX_LOOKS = {"none": 0, "golden": 1, "punchy": 2}


def tonemap(hdr: np.ndarray, look: Optional[str] = "none"):
    """AgX tonemap (reference core/tonemapping.hpp; look None = no tonemapper) + the 8-bit encoding of
    output/ppm.cpp of an (H, W, 4) float32 frame, on the device. Returns (ldr float32 (H,W,4), rgb8 (H,W,3))."""
    hdr = np.ascontiguousarray(hdr, np.float32)
    h, w = hdr.shape[:2]
    ldr = np.empty_like(hdr)
    rgb = np.empty((h, w, 3), np.uint8)
    _check(lib().yart_hip_tonemap_host(hdr.ctypes.data_as(C.c_void_p), w, h, -1 if look is None else AGX_LOOKS[look],
                                       ldr.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p)))
    return ldr, rgb


def write_ppm(path, rgb8: np.ndarray):
    """P6 file as output/ppm.cpp:10 writes it."""
    h, w = rgb8.shape[:2]
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb8, np.uint8).tobytes())
