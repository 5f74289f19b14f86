"""Scene description objects and the ``.yscn`` binary container.

The container stands where the output of the reference's glTF loader stands
(reference ``src/gltf/gltf.cpp:319-358``: materials, textures, meshes, a node
tree and the light list).  It is consumed by

* the product library (``yart_amd/csrc/scene_file.cpp`` or, through ctypes
  pointers, ``yart_amd/api.py``),
* the compiled reference (``oracle/ref_driver.cpp``) and the CPU restatement
  (``oracle/``), both through ``oracle/yscn.hpp``.

Byte layout: see ``oracle/yscn.hpp`` (kept in one place on purpose).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

MAGIC = b"YSCN0001"

TEX_LINEAR, TEX_SRGB, TEX_NONCOLOR = 0, 1, 2          # reference core/texture.hpp:15-19
LIGHT_AREA, LIGHT_UNIFORM_INF, LIGHT_IMAGE_INF = 0, 1, 2

IDENTITY = np.eye(4, dtype=np.float32)


@dataclass
class Texture:
    """8-bit (1-4 channel) or float RGB texture, in the in-memory form the
    reference holds after load (sRGB data already gamma-2 re-encoded,
    reference core/texture.hpp:78-84)."""
    data: np.ndarray                  # (h, w, c) uint8 or float32
    type: int = TEX_NONCOLOR

    @property
    def height(self): return self.data.shape[0]
    @property
    def width(self): return self.data.shape[1]
    @property
    def channels(self): return self.data.shape[2]
    @property
    def is_float(self): return self.data.dtype == np.float32


@dataclass
class Material:
    """Constructor arguments of the reference's ParametricBSDF
    (reference bsdf/parametric.hpp:16-37)."""
    base: tuple = (0.8, 0.8, 0.8)
    emission: tuple = (0.0, 0.0, 0.0)
    metallic: float = 0.0
    roughness: float = 0.0
    transmission: float = 0.0
    ior: float = 1.5
    anisotropic: float = 0.0
    aniso_rotation: float = 0.0
    clearcoat: float = 0.0
    clearcoat_roughness: float = 0.0
    normal_scale: float = 1.0
    thin_transmission: bool = False
    volume_color: tuple = (1.0, 1.0, 1.0)
    volume_density: float = 0.0
    tex_base: int = -1            # RGBA u8, sRGB
    tex_mr: int = -1              # 2ch u8 (roughness, metallic), NonColor
    tex_transmission: int = -1    # 1ch u8
    tex_normal: int = -1          # RGB u8, NonColor
    tex_clearcoat: int = -1       # 1ch u8
    tex_emission: int = -1        # RGB u8, sRGB

    def pack(self) -> bytes:
        return struct.pack(
            "<3f3f9fI3ff6i",
            *self.base, *self.emission,
            self.metallic, self.roughness, self.transmission, self.ior,
            self.anisotropic, self.aniso_rotation, self.clearcoat, self.clearcoat_roughness,
            self.normal_scale, int(self.thin_transmission), *self.volume_color,
            self.volume_density, self.tex_base, self.tex_mr, self.tex_transmission,
            self.tex_normal, self.tex_clearcoat, self.tex_emission)

    @property
    def is_emissive(self) -> bool:
        # reference parametric.cpp:67: m_hasEmission = length2(m_emission) > 0
        e = np.asarray(self.emission, dtype=np.float32)
        return float((e * e).sum()) > 0.0


@dataclass
class Mesh:
    positions: np.ndarray             # (nv, 3) f32
    normals: np.ndarray               # (nv, 3) f32
    tangents: np.ndarray              # (nv, 4) f32
    uvs: np.ndarray                   # (nv, 2) f32
    faces: np.ndarray                 # (nf, 4) u32: i0, i1, i2, material
    face_light: Optional[np.ndarray] = None   # (nf,) i32, -1 = not a light

    def __post_init__(self):
        self.positions = np.ascontiguousarray(self.positions, dtype=np.float32).reshape(-1, 3)
        nv = len(self.positions)
        self.normals = np.ascontiguousarray(self.normals, dtype=np.float32).reshape(nv, 3)
        if self.tangents is None:
            self.tangents = np.zeros((nv, 4), np.float32)
        self.tangents = np.ascontiguousarray(self.tangents, dtype=np.float32).reshape(nv, 4)
        if self.uvs is None:
            self.uvs = np.zeros((nv, 2), np.float32)
        self.uvs = np.ascontiguousarray(self.uvs, dtype=np.float32).reshape(nv, 2)
        self.faces = np.ascontiguousarray(self.faces, dtype=np.uint32).reshape(-1, 4)
        if self.face_light is None:
            self.face_light = np.full(len(self.faces), -1, np.int32)
        self.face_light = np.ascontiguousarray(self.face_light, dtype=np.int32)


@dataclass
class Node:
    parent: int = -1
    mesh: int = -1
    fwd: np.ndarray = field(default_factory=lambda: IDENTITY.copy())
    inv: np.ndarray = field(default_factory=lambda: IDENTITY.copy())


@dataclass
class Light:
    type: int = LIGHT_AREA
    mesh: int = -1
    tri: int = 0
    two_sided: bool = False
    texture: int = -1
    radius: float = 100.0
    emission: tuple = (0.0, 0.0, 0.0)
    fwd: np.ndarray = field(default_factory=lambda: IDENTITY.copy())
    inv: np.ndarray = field(default_factory=lambda: IDENTITY.copy())

    def pack(self) -> bytes:
        return (struct.pack("<IiIIif3f", self.type, self.mesh, self.tri, int(self.two_sided),
                            self.texture, self.radius, *self.emission)
                + np.asarray(self.fwd, np.float32).tobytes()
                + np.asarray(self.inv, np.float32).tobytes())


def trs(translation=(0, 0, 0), axis=(0, 1, 0), angle=0.0, scale=(1, 1, 1)):
    """float32 forward / inverse 4x4 pair (row-major, column vectors) for
    T * R * S — the order reference gltf.cpp:284-289 composes."""
    t = np.eye(4); t[:3, 3] = translation
    a = np.asarray(axis, np.float64); a = a / np.linalg.norm(a)
    c, s = np.cos(angle), np.sin(angle)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(4); R[:3, :3] = c * np.eye(3) + s * K + (1 - c) * np.outer(a, a)
    S = np.diag([*np.broadcast_to(np.asarray(scale, np.float64), 3), 1.0])
    m = t @ R @ S
    return m.astype(np.float32), np.linalg.inv(m).astype(np.float32)


@dataclass
class Scene:
    textures: List[Texture] = field(default_factory=list)
    materials: List[Material] = field(default_factory=list)
    meshes: List[Mesh] = field(default_factory=list)
    nodes: List[Node] = field(default_factory=lambda: [Node()])   # node 0 = root
    lights: List[Light] = field(default_factory=list)

    # -- construction helpers -------------------------------------------------
    def add_texture(self, tex: Texture) -> int:
        self.textures.append(tex); return len(self.textures) - 1

    def add_material(self, mat: Material) -> int:
        self.materials.append(mat); return len(self.materials) - 1

    def add_mesh(self, mesh: Mesh) -> int:
        self.meshes.append(mesh); return len(self.meshes) - 1

    def add_node(self, mesh: int = -1, parent: int = 0, fwd=None, inv=None) -> int:
        """Nodes must be appended in pre-order (parent before children)."""
        n = Node(parent=parent, mesh=mesh)
        if fwd is not None:
            n.fwd = np.asarray(fwd, np.float32); n.inv = np.asarray(inv, np.float32)
        self.nodes.append(n); return len(self.nodes) - 1

    def _global(self, idx: int):
        """node.transform * parentGlobal, composed the way reference
        gltf.cpp:293 does (child matrix on the LEFT), in float32 like mat.hpp:263-273."""
        chain = []
        while idx >= 0:
            chain.append(idx); idx = self.nodes[idx].parent
        fwd, inv = IDENTITY.copy(), IDENTITY.copy()
        for i in reversed(chain):          # root first
            n = self.nodes[i]
            fwd = _matmul32(n.fwd, fwd)
            inv = _matmul32(inv, n.inv)
        return fwd, inv

    def create_area_lights(self):
        """One AreaLight per emissive triangle, numbered per node, children before
        the node itself — reference gltf.cpp:295-314."""
        self.lights = [l for l in self.lights if l.type != LIGHT_AREA]
        area = []
        children = {i: [] for i in range(len(self.nodes))}
        for i, n in enumerate(self.nodes):
            if n.parent >= 0: children[n.parent].append(i)

        def visit(i):
            for c in children[i]: visit(c)
            n = self.nodes[i]
            if n.mesh < 0: return
            mesh = self.meshes[n.mesh]
            fwd, inv = self._global(i)
            li = 0
            for t in range(len(mesh.faces)):
                mat = self.materials[int(mesh.faces[t, 3])]
                if mat.is_emissive:
                    area.append(Light(LIGHT_AREA, mesh=n.mesh, tri=t, emission=tuple(mat.emission),
                                      fwd=fwd, inv=inv))
                    mesh.face_light[t] = li; li += 1
        visit(0)
        # area lights precede infinite lights (main.cpp adds env lights after load)
        self.lights = area + self.lights

    # -- I/O ------------------------------------------------------------------
    def tobytes(self) -> bytes:
        out = [MAGIC, struct.pack("<8I", len(self.textures), len(self.materials), len(self.meshes),
                                  len(self.nodes), len(self.lights), 0, 0, 0)]

        def arr(a):
            b = np.ascontiguousarray(a).tobytes()
            return b + b"\0" * ((4 - len(b) % 4) % 4)

        for t in self.textures:
            out.append(struct.pack("<5I", t.width, t.height, t.channels, int(t.is_float), t.type))
            out.append(arr(t.data))
        for m in self.materials:
            out.append(m.pack())
        for m in self.meshes:
            out.append(struct.pack("<2I", len(m.positions), len(m.faces)))
            out += [arr(m.positions), arr(m.normals), arr(m.tangents), arr(m.uvs), arr(m.faces),
                    arr(m.face_light)]
        for n in self.nodes:
            out.append(struct.pack("<2i", n.parent, n.mesh))
            out.append(np.asarray(n.fwd, np.float32).tobytes())
            out.append(np.asarray(n.inv, np.float32).tobytes())
        for l in self.lights:
            out.append(l.pack())
        return b"".join(out)

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.tobytes())

    @classmethod
    def frombytes(cls, buf: bytes) -> "Scene":
        """Inverse of :meth:`tobytes` (e.g. to inspect or edit what the glTF importer produced)."""
        if buf[:8] != MAGIC:
            raise ValueError("not a .yscn container")
        pos = 8
        nt, nm, nme, nn, nl, _, _, _ = struct.unpack_from("<8I", buf, pos); pos += 32

        def arr(dtype, count, shape):
            nonlocal pos
            a = np.frombuffer(buf, dtype=dtype, count=count, offset=pos).reshape(shape).copy()
            nbytes = a.nbytes
            pos += nbytes + (4 - nbytes % 4) % 4
            return a
        s = cls()
        for _ in range(nt):
            w, h, c, is_float, typ = struct.unpack_from("<5I", buf, pos); pos += 20
            s.textures.append(Texture(arr(np.float32 if is_float else np.uint8, w * h * c, (h, w, c)), typ))
        for _ in range(nm):
            v = struct.unpack_from("<3f3f9fI3ff6i", buf, pos); pos += 104
            s.materials.append(Material(base=v[0:3], emission=v[3:6], metallic=v[6], roughness=v[7], transmission=v[8],
                                        ior=v[9], anisotropic=v[10], aniso_rotation=v[11], clearcoat=v[12],
                                        clearcoat_roughness=v[13], normal_scale=v[14], thin_transmission=bool(v[15]),
                                        volume_color=v[16:19], volume_density=v[19], tex_base=v[20], tex_mr=v[21],
                                        tex_transmission=v[22], tex_normal=v[23], tex_clearcoat=v[24], tex_emission=v[25]))
        for _ in range(nme):
            nv, nf = struct.unpack_from("<2I", buf, pos); pos += 8
            p_ = arr(np.float32, nv * 3, (nv, 3)); n_ = arr(np.float32, nv * 3, (nv, 3))
            t_ = arr(np.float32, nv * 4, (nv, 4)); u_ = arr(np.float32, nv * 2, (nv, 2))
            f_ = arr(np.uint32, nf * 4, (nf, 4)); l_ = arr(np.int32, nf, (nf,))
            s.meshes.append(Mesh(p_, n_, t_, u_, f_, l_))
        s.nodes = []
        for _ in range(nn):
            parent, mesh = struct.unpack_from("<2i", buf, pos); pos += 8
            s.nodes.append(Node(parent, mesh, arr(np.float32, 16, (4, 4)), arr(np.float32, 16, (4, 4))))
        for _ in range(nl):
            typ, mesh, tri, two, tex, radius, e0, e1, e2 = struct.unpack_from("<IiIIif3f", buf, pos); pos += 36
            s.lights.append(Light(typ, mesh, tri, bool(two), tex, radius, (e0, e1, e2),
                                  arr(np.float32, 16, (4, 4)), arr(np.float32, 16, (4, 4))))
        return s

    @classmethod
    def load(cls, path) -> "Scene":
        with open(path, "rb") as f:
            return cls.frombytes(f.read())

    @property
    def n_triangles(self):
        return int(sum(len(self.meshes[n.mesh].faces) for n in self.nodes if n.mesh >= 0))


def _matmul32(a, b):
    """4x4 product accumulated in float32 in k order (reference mat.hpp:263-273)."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    r = np.zeros((4, 4), np.float32)
    for k in range(4):
        r = (r + np.outer(a[:, k], b[k, :]).astype(np.float32)).astype(np.float32)
    return r
