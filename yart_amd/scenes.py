"""Deterministic synthetic scene generators (no assets ship with the reference:
its ``models/`` and ``hdris/`` are git-ignored).

Each generator returns ``(Scene, params_dict)`` where ``params_dict`` holds the
camera / render settings of the BASELINE.json configuration it stands for, in
the vocabulary of ``oracle/params.hpp``.

* :func:`cornell`       — C1/C2: single-mesh Cornell box with a quad emitter.
* :func:`material_test` — small scene exercising every lobe, texture type, alpha
  cut-outs, normal maps, nested node transforms and an env map.
* :func:`heightfield`   — Cornell + 256x256 height field (131k triangles), the
  probe scene of BASELINE.md §2.
* :func:`sponza_class`  — C3/C4: env-lit atrium, ≈262k triangles, textured.
* :func:`mclaren_class` — C5: clearcoat / thin-glass / chrome body with DoF.
"""
from __future__ import annotations

import math

import numpy as np

from .yscn import (LIGHT_IMAGE_INF, LIGHT_UNIFORM_INF, TEX_LINEAR, TEX_NONCOLOR, TEX_SRGB, Light, Material, Mesh,
                   Scene, Texture, trs)


# --------------------------------------------------------------------------
# geometry helpers — every helper returns (positions, normals, tangents, uvs, tris)
# --------------------------------------------------------------------------
class MeshBuilder:
    def __init__(self):
        self.p, self.n, self.t, self.uv, self.f = [], [], [], [], []
        self.nv = 0

    def add(self, p, n, t, uv, tris, material):
        p = np.asarray(p, np.float32).reshape(-1, 3)
        k = len(p)
        self.p.append(p)
        self.n.append(np.broadcast_to(np.asarray(n, np.float32), (k, 3)).copy())
        if t is None:
            t = np.zeros((k, 4), np.float32)
        self.t.append(np.broadcast_to(np.asarray(t, np.float32), (k, 4)).copy())
        if uv is None:
            uv = np.zeros((k, 2), np.float32)
        self.uv.append(np.asarray(uv, np.float32).reshape(k, 2))
        tris = np.asarray(tris, np.uint32).reshape(-1, 3) + np.uint32(self.nv)
        mat = np.full((len(tris), 1), material, np.uint32)
        self.f.append(np.concatenate([tris, mat], axis=1))
        self.nv += k

    def quad(self, p0, p1, p2, p3, material, uv_scale=1.0):
        """p0..p3 counter-clockwise seen from the front; flat normal."""
        p = np.asarray([p0, p1, p2, p3], np.float64)
        e1, e2 = p[1] - p[0], p[3] - p[0]
        n = np.cross(e1, e2); n /= np.linalg.norm(n)
        tg = e1 / np.linalg.norm(e1)
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * uv_scale
        self.add(p, n, [*tg, 1.0], uv, [[0, 1, 2], [0, 2, 3]], material)

    def box(self, lo, hi, material, rot_y=0.0, centre=None):
        lo = np.asarray(lo, np.float64); hi = np.asarray(hi, np.float64)
        c = (lo + hi) / 2 if centre is None else np.asarray(centre, np.float64)
        h = (hi - lo) / 2
        cs, sn = math.cos(rot_y), math.sin(rot_y)
        R = np.array([[cs, 0, sn], [0, 1, 0], [-sn, 0, cs]])

        def P(x, y, z):
            return R @ (np.array([x, y, z]) * h) + c
        faces = [
            [P(-1, -1, 1), P(1, -1, 1), P(1, 1, 1), P(-1, 1, 1)],      # +z
            [P(1, -1, -1), P(-1, -1, -1), P(-1, 1, -1), P(1, 1, -1)],  # -z
            [P(1, -1, 1), P(1, -1, -1), P(1, 1, -1), P(1, 1, 1)],      # +x
            [P(-1, -1, -1), P(-1, -1, 1), P(-1, 1, 1), P(-1, 1, -1)],  # -x
            [P(-1, 1, 1), P(1, 1, 1), P(1, 1, -1), P(-1, 1, -1)],      # +y
            [P(-1, -1, -1), P(1, -1, -1), P(1, -1, 1), P(-1, -1, 1)],  # -y
        ]
        for q in faces:
            self.quad(*q, material)

    def grid(self, fn, nu, nv, material, uv_scale=(1.0, 1.0), flip=False):
        """Parametric surface fn(u, v) -> (x, y, z) on an (nu+1) x (nv+1) vertex
        lattice with smooth normals/tangents from finite differences."""
        u = np.linspace(0.0, 1.0, nu + 1); v = np.linspace(0.0, 1.0, nv + 1)
        U, V = np.meshgrid(u, v, indexing="xy")            # (nv+1, nu+1)
        P = np.stack(fn(U, V), axis=-1).astype(np.float64)
        eps = 1e-4
        Pu = (np.stack(fn(np.clip(U + eps, 0, 1), V), -1) - np.stack(fn(np.clip(U - eps, 0, 1), V), -1))
        Pv = (np.stack(fn(U, np.clip(V + eps, 0, 1)), -1) - np.stack(fn(U, np.clip(V - eps, 0, 1)), -1))
        N = np.cross(Pu, Pv)
        if flip: N = -N
        ln = np.linalg.norm(N, axis=-1, keepdims=True); ln[ln == 0] = 1
        N = N / ln
        T = Pu / np.maximum(np.linalg.norm(Pu, axis=-1, keepdims=True), 1e-20)
        T4 = np.concatenate([T, np.ones_like(T[..., :1])], -1)
        UV = np.stack([U * uv_scale[0], V * uv_scale[1]], -1)
        idx = np.arange((nu + 1) * (nv + 1)).reshape(nv + 1, nu + 1)
        a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1]
        if flip:
            tris = np.stack([np.stack([a, c, b], -1), np.stack([a, d, c], -1)], -2)
        else:
            tris = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], -2)
        self.add(P.reshape(-1, 3), N.reshape(-1, 3), T4.reshape(-1, 4), UV.reshape(-1, 2),
                 tris.reshape(-1, 3), material)

    def sphere(self, centre, radius, material, nu=32, nv=16, uv_scale=(1.0, 1.0)):
        cx, cy, cz = centre

        def fn(U, V):
            phi = U * 2 * np.pi; th = V * np.pi
            return (cx + radius * np.sin(th) * np.cos(phi), cy + radius * np.cos(th),
                    cz + radius * np.sin(th) * np.sin(phi))
        # analytic normals/tangents (finite differences degenerate at the poles)
        u = np.linspace(0.0, 1.0, nu + 1); v = np.linspace(0.0, 1.0, nv + 1)
        U, V = np.meshgrid(u, v, indexing="xy")
        phi = U * 2 * np.pi; th = V * np.pi
        N = np.stack([np.sin(th) * np.cos(phi), np.cos(th), np.sin(th) * np.sin(phi)], -1)
        P = np.asarray(centre) + radius * N
        T = np.stack([-np.sin(phi), np.zeros_like(phi), np.cos(phi)], -1)
        T4 = np.concatenate([T, np.ones_like(T[..., :1])], -1)
        UV = np.stack([U * uv_scale[0], V * uv_scale[1]], -1)
        idx = np.arange((nu + 1) * (nv + 1)).reshape(nv + 1, nu + 1)
        a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1]
        tris = np.stack([np.stack([a, c, b], -1), np.stack([a, d, c], -1)], -2)
        self.add(P.reshape(-1, 3), N.reshape(-1, 3), T4.reshape(-1, 4), UV.reshape(-1, 2),
                 tris.reshape(-1, 3), material)

    def build(self) -> Mesh:
        return Mesh(np.concatenate(self.p), np.concatenate(self.n), np.concatenate(self.t),
                    np.concatenate(self.uv), np.concatenate(self.f))


# --------------------------------------------------------------------------
# procedural textures (stored in the post-load form the reference keeps)
# --------------------------------------------------------------------------
def _srgb_store(linear):
    """linear [0,1] -> byte holding sqrt(linear) (gamma-2), as texture.hpp:78-84 re-encodes."""
    return np.clip(np.sqrt(np.clip(linear, 0, 1)) * 255.0, 0, 255).astype(np.uint8)


def _noise(size, seed, octaves=4):
    rng = np.random.RandomState(seed)
    out = np.zeros((size, size), np.float64)
    for o in range(octaves):
        n = 4 << o
        g = rng.rand(n, n)
        reps = size // n
        up = np.kron(g, np.ones((reps, reps)))
        # cheap smoothing: average with rolled copies
        up = (up + np.roll(up, reps // 2, 0) + np.roll(up, reps // 2, 1)) / 3
        out += up / (1 << o)
    out -= out.min(); out /= out.max()
    return out


def tex_base_color(size, seed, c0, c1, alpha=None):
    n = _noise(size, seed)
    yy, xx = np.mgrid[0:size, 0:size]
    brick = (((yy // (size // 16)) + (xx // (size // 8))) % 2) * 0.25
    m = np.clip(n * 0.75 + brick, 0, 1)[..., None]
    rgb = np.asarray(c0) * (1 - m) + np.asarray(c1) * m
    a = np.full((size, size, 1), 255, np.uint8) if alpha is None else alpha[..., None]
    return Texture(np.concatenate([_srgb_store(rgb), a], -1), TEX_SRGB)


def tex_alpha_leaves(size, seed):
    """alpha cut-out pattern (foliage-like): discs on a lattice + noise."""
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    cell = size / 8
    dx = (xx % cell) - cell / 2; dy = (yy % cell) - cell / 2
    r = np.sqrt(dx * dx + dy * dy) / (cell / 2)
    n = _noise(size, seed)
    a = np.clip((0.95 - r + 0.3 * (n - 0.5)) * 6.0, 0, 1)
    return (a * 255).astype(np.uint8)


def tex_metal_rough(size, seed, rough=(0.3, 0.9), metal=(0.0, 0.0)):
    n = _noise(size, seed)
    r = rough[0] + (rough[1] - rough[0]) * n
    m = metal[0] + (metal[1] - metal[0]) * (n > 0.5)
    return Texture(np.stack([np.clip(r * 255, 0, 255), np.clip(m * 255, 0, 255)], -1).astype(np.uint8),
                   TEX_NONCOLOR)


def tex_normal(size, seed, strength=0.6):
    h = _noise(size, seed)
    gx = (np.roll(h, -1, 1) - np.roll(h, 1, 1)) * strength * size / 64
    gy = (np.roll(h, -1, 0) - np.roll(h, 1, 0)) * strength * size / 64
    n = np.stack([-gx, -gy, np.ones_like(h)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return Texture(np.clip((n * 0.5 + 0.5) * 255, 0, 255).astype(np.uint8), TEX_NONCOLOR)


def tex_mono(size, seed, lo=0.0, hi=1.0):
    n = _noise(size, seed)
    return Texture(np.clip((lo + (hi - lo) * n) * 255, 0, 255).astype(np.uint8)[..., None], TEX_NONCOLOR)


def sky_octahedral(size, sun_dir=(0.35, 0.8, 0.25), sun_power=60.0, turbidity=1.0):
    """Procedural sky + sun in the octahedral layout ImageInfiniteLight expects
    (reference math.hpp:151-179 / light.cpp:137-238)."""
    v, u = np.mgrid[0:size, 0:size].astype(np.float64)
    u = (u + 0.5) / size; v = (v + 0.5) / size
    x = 2 * u - 1; z = 2 * v - 1
    y = 1 - (np.abs(x) + np.abs(z))
    neg = y < 0
    xo = x.copy()
    x = np.where(neg, (1 - np.abs(z)) * np.sign(x + 1e-30), x)
    z = np.where(neg, (1 - np.abs(xo)) * np.sign(z + 1e-30), z)
    d = np.stack([x, y, z], -1); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    s = np.asarray(sun_dir, np.float64); s /= np.linalg.norm(s)
    up = np.clip(d[..., 1], -1, 1)
    horizon = np.exp(-np.abs(up) * 4.0)
    sky = (np.array([0.25, 0.45, 0.95]) * (0.35 + 0.65 * np.clip(up, 0, 1))[..., None]
           + np.array([0.9, 0.85, 0.75]) * horizon[..., None] * 0.6) * turbidity
    ground = np.array([0.18, 0.16, 0.14]) * (0.3 + 0.2 * horizon)[..., None]
    rgb = np.where((up > 0)[..., None], sky, ground)
    cs = np.clip((d * s).sum(-1), -1, 1)
    sun = np.exp((cs - 1.0) * 900.0) * sun_power + np.exp((cs - 1.0) * 30.0) * 1.5
    rgb = rgb + sun[..., None] * np.array([1.0, 0.93, 0.8])
    return Texture(rgb.astype(np.float32), TEX_LINEAR)


# --------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------
def _white(**kw):
    return Material(base=(0.73, 0.73, 0.73), roughness=1.0, **kw)


def cornell(width=256, height=256, spp=16, depth=4):
    """BASELINE configs[0]/[1]: Cornell-style single mesh, camera preset of
    reference main.cpp:36."""
    s = Scene()
    white = s.add_material(_white())
    red = s.add_material(Material(base=(0.65, 0.05, 0.05), roughness=1.0))
    green = s.add_material(Material(base=(0.12, 0.45, 0.15), roughness=1.0))
    light = s.add_material(Material(base=(0.78, 0.78, 0.78), roughness=1.0, emission=(15.0, 15.0, 15.0)))
    b = MeshBuilder()
    L, H = 5.0, 10.0
    b.quad((-L, 0, L), (L, 0, L), (L, 0, -L), (-L, 0, -L), white)          # floor (+y)
    b.quad((-L, H, -L), (L, H, -L), (L, H, L), (-L, H, L), white)          # ceiling (-y)
    b.quad((-L, 0, -L), (L, 0, -L), (L, H, -L), (-L, H, -L), white)        # back (+z)
    b.quad((-L, 0, L), (-L, 0, -L), (-L, H, -L), (-L, H, L), red)          # left (+x)
    b.quad((L, 0, -L), (L, 0, L), (L, H, L), (L, H, -L), green)            # right (-x)
    e = 1.5
    b.quad((-e, H - 0.01, -e), (e, H - 0.01, -e), (e, H - 0.01, e), (-e, H - 0.01, e), light)
    b.box((-3.2, 0, -3.4), (-0.2, 6.0, -0.4), white, rot_y=0.3)
    b.box((0.6, 0, 0.2), (3.6, 3.0, 3.2), white, rot_y=-0.31)
    s.add_node(s.add_mesh(b.build()))
    s.create_area_lights()
    p = dict(size=(width, height), spp=spp, depth=depth, focal=35.0, fnumber=0.0,
             eye=(0.0, 5.0, 15.0), target=(0.0, 5.0, 0.0), up=(0.0, 1.0, 0.0), exposure=0.0,
             background=(0.0, 0.0, 0.0))
    return s, p


def uniform_sky(width=64, height=64, spp=16, depth=4, emission=(0.6, 0.7, 1.0)):
    """The Cornell box (open towards the camera) under the reference's alternative environment preset, a
    UniformInfiniteLight (reference main.cpp:86, core/light.cpp:83-131): escaping rays pick up its emission, and the
    power light sampler spends half of its NEE choices on a light whose sample() returns nothing."""
    s, p = cornell(width, height, spp, depth)
    s.lights.append(Light(LIGHT_UNIFORM_INF, radius=100.0, emission=tuple(emission)))
    return s, p


def two_skies(width=48, height=32, spp=8, depth=5, tex=32):
    """material_test with a UniformInfiniteLight NEXT TO its image light: two infinite lights at once
    (mis-integrator.cpp:27-43 loops over all of them on a miss; light-sampler.cpp:52-78 counts both in pInfinite)."""
    s, p = material_test(width, height, spp, depth, tex=tex)
    s.lights.append(Light(LIGHT_UNIFORM_INF, radius=100.0, emission=(0.3, 0.25, 0.2)))
    return s, p


def material_test(width=192, height=128, spp=16, depth=6, tex=64):
    """Every lobe / texture type / quirk in one small scene: metallic (rough,
    smooth, anisotropic), rough + smooth dielectric (thin and refractive with
    volume absorption), glossy-diffuse with all texture kinds, clearcoat,
    emissive-textured panel, alpha cut-out quad in front of a light, nested node
    transforms (scaled, rotated instance of a shared mesh), an area light and an
    env map, DoF with a polygonal aperture."""
    s = Scene()
    t_base = s.add_texture(tex_base_color(tex, 1, (0.7, 0.25, 0.15), (0.9, 0.8, 0.6)))
    t_leaf = s.add_texture(tex_base_color(tex, 2, (0.1, 0.4, 0.1), (0.3, 0.6, 0.2),
                                          alpha=tex_alpha_leaves(tex, 3)))
    t_mr = s.add_texture(tex_metal_rough(tex, 4, (0.15, 0.8), (0.0, 1.0)))
    t_nrm = s.add_texture(tex_normal(tex, 5))
    t_tr = s.add_texture(tex_mono(tex, 6, 0.2, 1.0))
    t_cc = s.add_texture(tex_mono(tex, 7, 0.3, 1.0))
    t_em = s.add_texture(Texture(_srgb_store(np.stack([_noise(tex, 8)] * 3, -1)), TEX_SRGB))
    t_sky = s.add_texture(sky_octahedral(32, sun_power=20.0))

    M = s.add_material
    floor = M(Material(base=(1, 1, 1), roughness=1.0, tex_base=t_base, tex_mr=t_mr, tex_normal=t_nrm))
    mats = [
        M(Material(base=(0.95, 0.64, 0.54), metallic=1.0, roughness=0.35)),                 # copper
        M(Material(base=(0.9, 0.9, 0.9), metallic=1.0, roughness=0.0)),                     # mirror
        M(Material(base=(1.0, 0.78, 0.34), metallic=1.0, roughness=0.4, anisotropic=0.8,
                   aniso_rotation=0.6)),                                                    # brushed gold
        M(Material(base=(0.9, 0.95, 1.0), transmission=1.0, roughness=0.25, ior=1.45,
                   thin_transmission=True)),                                                # thin frosted
        M(Material(base=(1.0, 1.0, 1.0), transmission=1.0, roughness=0.0, ior=1.5,
                   volume_color=(0.6, 0.9, 0.7), volume_density=0.8)),                      # solid glass
        M(Material(base=(0.9, 0.9, 1.0), transmission=0.9, roughness=0.3, ior=1.33,
                   tex_transmission=t_tr)),                                                 # rough refractive
        M(Material(base=(0.8, 0.1, 0.1), roughness=0.5, clearcoat=1.0, clearcoat_roughness=0.03)),
        M(Material(base=(0.1, 0.2, 0.8), roughness=0.6, metallic=0.5, clearcoat=0.8,
                   clearcoat_roughness=0.2, tex_clearcoat=t_cc)),
        M(Material(base=(0.8, 0.8, 0.8), roughness=0.0)),                                   # smooth plastic
        M(Material(base=(1, 1, 1), roughness=0.7, tex_base=t_base, tex_mr=t_mr)),
    ]
    leaf = M(Material(base=(1, 1, 1), roughness=0.8, tex_base=t_leaf))
    glow = M(Material(base=(0.5, 0.5, 0.5), roughness=1.0, emission=(4.0, 3.0, 2.0), tex_emission=t_em))
    lamp = M(Material(base=(0.8, 0.8, 0.8), roughness=1.0, emission=(20.0, 20.0, 20.0)))

    b = MeshBuilder()
    b.grid(lambda U, V: ((U - 0.5) * 16, 0 * U, (V - 0.5) * 12), 8, 6, floor, uv_scale=(4, 3), flip=True)
    for i, m in enumerate(mats):
        x = -6.0 + 1.35 * i + (0.3 if i % 2 else 0.0)
        z = -1.5 + 2.0 * (i % 3)
        b.sphere((x, 0.8, z), 0.8, m, nu=20, nv=10, uv_scale=(2, 1))
    b.quad((-2, 0.2, 3.0), (2, 0.2, 3.0), (2, 2.4, 3.0), (-2, 2.4, 3.0), leaf, uv_scale=2.0)   # alpha card
    b.quad((-7.5, 0.5, -5.0), (-4.5, 0.5, -5.0), (-4.5, 2.5, -5.0), (-7.5, 2.5, -5.0), glow)
    b.quad((-1, 5.0, -1), (1, 5.0, -1), (1, 5.0, 1), (-1, 5.0, 1), lamp)
    main = s.add_mesh(b.build())
    s.add_node(main)

    # shared instanced mesh under nested transforms (exercises testNode recursion and
    # the per-level normal renormalisation, ray-integrator.cpp:26-29, 50-52)
    bb = MeshBuilder()
    bb.box((-0.5, 0, -0.5), (0.5, 1, 0.5), mats[0])
    bb.sphere((0, 1.4, 0), 0.4, mats[6], nu=12, nv=6)
    inst = s.add_mesh(bb.build())
    g = s.add_node(-1, 0, *trs((5.0, 0.0, -3.0), (0, 1, 0), 0.5, (1.0, 1.0, 1.0)))
    s.add_node(inst, g, *trs((0, 0, 0), (0, 1, 0), 0.0, (1.0, 1.5, 1.0)))
    s.add_node(inst, g, *trs((1.6, 0, 1.0), (0, 0, 1), 0.35, (0.7, 0.7, 0.7)))
    s.create_area_lights()
    s.lights.append(Light(LIGHT_IMAGE_INF, texture=t_sky, radius=100.0))
    p = dict(size=(width, height), spp=spp, depth=depth, focal=35.0, fnumber=2.8, aperture_sides=6,
             eye=(1.0, 4.0, 11.0), target=(0.0, 0.8, 0.0), up=(0.0, 1.0, 0.0), exposure=0.5,
             background=(0.02, 0.02, 0.03))
    return s, p


def heightfield(width=256, height=256, spp=16, depth=4, n=256, env=False):
    """BASELINE.md §2 probe scene: Cornell + n x n height field (2 n^2 triangles)."""
    s, p = cornell(width, height, spp, depth)
    white = 0
    b = MeshBuilder()

    def fn(U, V):
        x = (U - 0.5) * 9.0; z = (V - 0.5) * 9.0
        y = 0.6 + 0.5 * np.sin(3.1 * x) * np.cos(2.3 * z) + 0.15 * np.sin(11.0 * x + 5.0 * z)
        return x, y, z
    b.grid(fn, n, n, white, flip=True)
    s.add_node(s.add_mesh(b.build()))
    s.create_area_lights()
    if env:
        t = s.add_texture(sky_octahedral(64))
        s.lights.append(Light(LIGHT_IMAGE_INF, texture=t, radius=100.0))
        p["depth"] = 8
    return s, p


def sponza_class(width=1920, height=1080, spp=256, depth=8, detail=1.0, tex=1024, sky=2048):
    """BASELINE configs[2]/[3]: env-lit two-storey colonnaded atrium, ≈262k
    triangles at detail=1, 24 materials with procedural base-colour / metal-rough /
    normal textures, alpha cut-out foliage, cloth banners; no area lights; camera
    scaled from the reference's "Sponza 2" preset (main.cpp:69-72), 35 mm f/4,
    exposure +5 EV (main.cpp:32-34) compensated by a dim sky."""
    s = Scene()
    rng = np.random.RandomState(1)
    d = max(detail, 0.05)

    def T(t): return s.add_texture(t)
    stone_tex = [T(tex_base_color(tex, 10 + i, c0, c1)) for i, (c0, c1) in enumerate([
        ((0.55, 0.5, 0.42), (0.8, 0.75, 0.65)), ((0.45, 0.4, 0.36), (0.7, 0.62, 0.5)),
        ((0.6, 0.55, 0.5), (0.85, 0.82, 0.78)), ((0.35, 0.3, 0.28), (0.6, 0.5, 0.42))])]
    mr_tex = [T(tex_metal_rough(tex, 20 + i, (0.45 + 0.1 * i, 0.95))) for i in range(3)]
    nrm_tex = [T(tex_normal(tex, 30 + i, 0.4 + 0.2 * i)) for i in range(3)]
    cloth_tex = [T(tex_base_color(tex, 40 + i, c0, c1)) for i, (c0, c1) in enumerate([
        ((0.6, 0.05, 0.05), (0.8, 0.2, 0.1)), ((0.05, 0.15, 0.5), (0.1, 0.3, 0.7)),
        ((0.1, 0.4, 0.1), (0.3, 0.55, 0.2))])]
    leaf_tex = T(tex_base_color(tex, 50, (0.08, 0.3, 0.06), (0.3, 0.55, 0.15),
                                alpha=tex_alpha_leaves(tex, 51)))
    M = s.add_material
    stone = [M(Material(base=(1, 1, 1), roughness=1.0, tex_base=stone_tex[i % 4], tex_mr=mr_tex[i % 3],
                        tex_normal=nrm_tex[i % 3])) for i in range(8)]
    plain = [M(Material(base=tuple(0.4 + 0.5 * rng.rand(3)), roughness=0.5 + 0.5 * rng.rand()))
             for _ in range(6)]
    cloth = [M(Material(base=(1, 1, 1), roughness=0.9, tex_base=cloth_tex[i])) for i in range(3)]
    leaf = M(Material(base=(1, 1, 1), roughness=0.7, tex_base=leaf_tex))
    bronze = M(Material(base=(0.8, 0.5, 0.25), metallic=1.0, roughness=0.4))
    tiles = [M(Material(base=(0.9, 0.9, 0.9), roughness=0.15, clearcoat=0.5, clearcoat_roughness=0.05,
                        tex_base=stone_tex[2])),
             M(Material(base=(0.3, 0.3, 0.32), roughness=0.25, tex_base=stone_tex[3]))]
    water = M(Material(base=(0.8, 0.9, 1.0), transmission=1.0, roughness=0.0, ior=1.33,
                       thin_transmission=True))

    b = MeshBuilder()
    LX, LZ, H1, H2 = 14.0, 6.0, 5.0, 10.0      # half-length, half-width, storey heights
    k = lambda v: max(2, int(round(v * math.sqrt(d))))

    # floor tiles (two alternating materials) and ground outside
    nx, nz = 28, 12
    for i in range(nx):
        for j in range(nz):
            x0 = -LX + 2 * LX * i / nx; x1 = -LX + 2 * LX * (i + 1) / nx
            z0 = -LZ + 2 * LZ * j / nz; z1 = -LZ + 2 * LZ * (j + 1) / nz
            b.quad((x0, 0, z1), (x1, 0, z1), (x1, 0, z0), (x0, 0, z0), tiles[(i + j) % 2])

    # outer walls (long sides + short ends), slightly displaced grids so they carry geometry
    def wall(x0, z0, x1, z1, y0, y1, mat, nu, nv, amp=0.03, flip=False):
        nrm = np.array([-(z1 - z0), 0.0, (x1 - x0)]); nrm /= np.linalg.norm(nrm)

        def fn(U, V):
            x = x0 + (x1 - x0) * U; z = z0 + (z1 - z0) * U; y = y0 + (y1 - y0) * V
            dsp = amp * (np.sin(37 * U * (abs(x1 - x0) + abs(z1 - z0))) * np.sin(29 * V * (y1 - y0)))
            return x + nrm[0] * dsp, y, z + nrm[2] * dsp
        b.grid(fn, nu, nv, mat, uv_scale=((abs(x1 - x0) + abs(z1 - z0)) / 4, (y1 - y0) / 4), flip=flip)
    W = LZ + 3.0
    wall(-LX - 3, -W, LX + 3, -W, 0, H2 + 2, stone[0], k(220), k(70))
    wall(LX + 3, W, -LX - 3, W, 0, H2 + 2, stone[1], k(220), k(70))
    wall(-LX - 3, W, -LX - 3, -W, 0, H2 + 2, stone[2], k(90), k(70))
    wall(LX + 3, -W, LX + 3, W, 0, H2 + 2, stone[3], k(90), k(70))

    # colonnade: columns (fluted cylinders) + arches, two storeys, both sides
    ncol = 10
    for side in (-1, 1):
        for storey, (y0, y1) in enumerate(((0.0, H1 - 1.2), (H1 + 0.3, H2 - 1.2))):
            for c in range(ncol + 1):
                cx = -LX + 2 * LX * c / ncol; cz = side * LZ
                rad = 0.38 if storey == 0 else 0.3
                mat = stone[4 + (c + storey) % 4]

                def col(U, V, cx=cx, cz=cz, rad=rad, y0=y0, y1=y1):
                    phi = U * 2 * np.pi
                    r = rad * (1.0 + 0.06 * np.cos(12 * phi)) * (1.0 - 0.12 * V + 0.1 * np.exp(-30 * V)
                                                                + 0.12 * np.exp(-30 * (1 - V)))
                    return cx + r * np.cos(phi), y0 + (y1 - y0) * V, cz + r * np.sin(phi)
                b.grid(col, k(36), k(30), mat, uv_scale=(2, 4), flip=True)
                b.box((cx - 0.5, y0, cz - 0.5), (cx + 0.5, y0 + 0.25, cz + 0.5), mat)
                b.box((cx - 0.5, y1, cz - 0.5), (cx + 0.5, y1 + 0.2, cz + 0.5), mat)
            for c in range(ncol):
                xa = -LX + 2 * LX * c / ncol; xb = -LX + 2 * LX * (c + 1) / ncol
                cz = side * LZ
                yb = y1 + 0.2

                def arch(U, V, xa=xa, xb=xb, cz=cz, yb=yb, side=side):
                    th = np.pi * U
                    xm = (xa + xb) / 2; rx = (xb - xa) / 2 - 0.3
                    return (xm - rx * np.cos(th), yb + 1.0 * np.sin(th),
                            cz + side * (-0.35 + 0.7 * V))
                b.grid(arch, k(40), k(6), stone[(c + 2) % 4], uv_scale=(3, 0.5), flip=(side < 0))
                # spandrel wall above the arch
                wall(xa, cz - side * 0.35, xb, cz - side * 0.35, yb + 1.0, yb + 1.2 + 0.25,
                     stone[(c + 1) % 4], k(16), k(3), amp=0.0, flip=(side > 0))
        # gallery floor slabs between the wall and the colonnade
        za, zb = sorted((side * LZ, side * (LZ + 3.0)))
        b.box((-LX - 3, H1 - 0.3, za), (LX + 3, H1, zb), plain[1])
        b.box((-LX - 3, H2 - 0.3, za), (LX + 3, H2, zb), plain[2])

    # cloth banners hanging across the atrium (tessellated, wavy)
    nb = 5
    for i in range(nb):
        x = -11.0 + 14.0 * i / (nb - 1)

        def ban(U, V, x=x, i=i):
            z = (U - 0.5) * 2 * (LZ - 0.6)
            sag = 1.2 * (1 - (2 * U - 1) ** 2)
            y = H2 + 0.6 - sag - 2.0 * V
            xx = x + 0.25 * np.sin(6 * U * np.pi + i) * (0.3 + V) + 0.1 * np.sin(9 * V + i)
            return xx, y, z
        b.grid(ban, k(70), k(36), cloth[i % 3], uv_scale=(4, 1))

    # ivy / foliage cards with alpha cut-outs on the long walls and columns
    nl = int(260 * d)
    for i in range(nl):
        side = -1 if i % 2 else 1
        x = -LX + 2 * LX * rng.rand(); y = 0.5 + (H2 - 1.5) * rng.rand()
        z = side * (W - 0.15 - 0.5 * rng.rand())
        w, h = 0.8 + 0.8 * rng.rand(), 0.8 + 0.8 * rng.rand()
        tilt = 0.3 * (rng.rand() - 0.5)
        if side > 0:
            b.quad((x + w, y, z + tilt), (x, y, z - tilt), (x, y + h, z - tilt), (x + w, y + h, z + tilt),
                   leaf, uv_scale=2.0)
        else:
            b.quad((x, y, z - tilt), (x + w, y, z + tilt), (x + w, y + h, z + tilt), (x, y + h, z - tilt),
                   leaf, uv_scale=2.0)

    # central fountain: basin ring, thin water sheet, bronze sphere
    def basin(U, V):
        phi = U * 2 * np.pi
        r = 1.6 + 0.5 * np.sin(np.pi * V)
        return r * np.cos(phi), 0.05 + 0.9 * V, r * np.sin(phi)
    b.grid(basin, k(64), k(12), stone[2], uv_scale=(6, 1), flip=True)
    b.grid(lambda U, V: ((U - 0.5) * 3.0, 0.75 + 0.02 * np.sin(20 * U) * np.cos(17 * V), (V - 0.5) * 3.0),
           k(40), k(40), water, flip=True)
    b.sphere((0, 2.0, 0), 0.7, bronze, nu=k(64), nv=k(32))
    # roof frame: beams leaving the atrium open to the sky
    for i in range(9):
        x = -LX + 2 * LX * i / 8
        b.box((x - 0.15, H2 + 1.2, -W), (x + 0.15, H2 + 1.6, W), plain[3])
    main = s.add_mesh(b.build())
    s.add_node(main)

    # instanced urns (shared mesh, per-node transforms) along the galleries
    ub = MeshBuilder()

    def urn(U, V):
        phi = U * 2 * np.pi
        r = 0.18 + 0.22 * np.sin(np.pi * V) ** 1.5 + 0.05 * np.cos(8 * phi) * np.sin(np.pi * V)
        return r * np.cos(phi), 0.9 * V, r * np.sin(phi)
    ub.grid(urn, k(40), k(24), bronze, flip=True)
    um = s.add_mesh(ub.build())
    grp = s.add_node(-1, 0)
    for i in range(12):
        side = -1 if i % 2 else 1
        x = -LX + 2.0 + (2 * LX - 4.0) * (i // 2) / 5
        s.add_node(um, grp, *trs((x, H1, side * (LZ + 1.2)), (0, 1, 0), 0.37 * i,
                                 (1.0 + 0.1 * (i % 3),) * 3))

    s.create_area_lights()
    t_sky = s.add_texture(sky_octahedral(sky, sun_dir=(0.25, 0.85, 0.35), sun_power=2.2, turbidity=0.035))
    s.lights.append(Light(LIGHT_IMAGE_INF, texture=t_sky, radius=100.0))
    p = dict(size=(width, height), spp=spp, depth=depth, focal=35.0, fnumber=4.0,
             eye=(11.14, 7.02, -1.25), target=(-8.05, 6.0, 0.04), up=(0.0, 1.0, 0.0), exposure=5.0,
             background=(0.0, 0.0, 0.0))
    return s, p


def mclaren_class(width=3840, height=2160, spp=512, depth=8, detail=1.0, tex=1024, sky=2048):
    """BASELINE configs[4]: car-like body (superellipsoid panels, wheels, glass
    canopy) on a ground plane; paint = base + clearcoat, thin glass (the reference
    loader forces thin transmission, gltf.cpp:105), smooth chrome; env-lit; f/2.8."""
    s = Scene()
    d = max(detail, 0.02)
    k = lambda v: max(3, int(round(v * math.sqrt(d))))
    M = s.add_material
    t_ground = s.add_texture(tex_base_color(tex, 60, (0.25, 0.25, 0.27), (0.4, 0.4, 0.42)))
    t_mr = s.add_texture(tex_metal_rough(tex, 61, (0.35, 0.8)))
    t_flake = s.add_texture(tex_metal_rough(tex, 62, (0.25, 0.45), (0.0, 1.0)))
    paint = M(Material(base=(0.75, 0.04, 0.03), roughness=0.45, metallic=0.6, clearcoat=1.0,
                       clearcoat_roughness=0.03, tex_mr=t_flake))
    glass = M(Material(base=(0.92, 0.96, 1.0), transmission=1.0, roughness=0.0, ior=1.5,
                       thin_transmission=True))
    chrome = M(Material(base=(0.95, 0.95, 0.97), metallic=1.0, roughness=0.0))
    rubber = M(Material(base=(0.03, 0.03, 0.03), roughness=0.85))
    carbon = M(Material(base=(0.06, 0.06, 0.07), roughness=0.35, clearcoat=0.6, clearcoat_roughness=0.1))
    ground = M(Material(base=(1, 1, 1), roughness=1.0, tex_base=t_ground, tex_mr=t_mr))
    b = MeshBuilder()
    b.grid(lambda U, V: ((U - 0.5) * 60, 0 * U, (V - 0.5) * 60), k(200), k(200), ground,
           uv_scale=(20, 20), flip=True)

    def superell(cx, cy, cz, rx, ry, rz, e1, e2, mat, nu, nv):
        def fn(U, V):
            phi = (U - 0.5) * 2 * np.pi; th = (V - 0.5) * np.pi
            f = lambda w, e: np.sign(w) * np.abs(w) ** e
            return (cx + rx * f(np.cos(th), e1) * f(np.cos(phi), e2),
                    cy + ry * f(np.sin(th), e1),
                    cz + rz * f(np.cos(th), e1) * f(np.sin(phi), e2))
        b.grid(fn, nu, nv, mat, uv_scale=(4, 2))
    superell(0, 0.55, 0, 2.3, 0.38, 0.95, 0.5, 0.6, paint, k(760), k(380))        # body
    superell(-0.2, 0.95, 0, 1.0, 0.33, 0.7, 0.7, 0.8, glass, k(360), k(180))      # canopy
    superell(1.9, 0.4, 0, 0.5, 0.12, 1.0, 0.3, 0.4, carbon, k(200), k(100))       # splitter
    superell(-2.2, 0.95, 0, 0.25, 0.05, 0.9, 0.3, 0.3, carbon, k(200), k(100))    # wing
    for sx in (-1.45, 1.45):
        for sz in (-0.95, 0.95):
            def tyre(U, V, sx=sx, sz=sz):
                phi = U * 2 * np.pi; th = V * 2 * np.pi
                R, r = 0.27, 0.11
                return (sx + (R + r * np.cos(th)) * np.cos(phi), 0.38 + (R + r * np.cos(th)) * np.sin(phi),
                        sz + r * 1.3 * np.sin(th))
            b.grid(tyre, k(200), k(80), rubber)
            superell(sx, 0.38, sz, 0.2, 0.2, 0.09, 1.0, 1.0, chrome, k(120), k(60))
    s.add_node(s.add_mesh(b.build()))
    s.create_area_lights()
    t_sky = s.add_texture(sky_octahedral(sky, sun_dir=(-0.4, 0.6, 0.5), sun_power=40.0))
    s.lights.append(Light(LIGHT_IMAGE_INF, texture=t_sky, radius=100.0))
    p = dict(size=(width, height), spp=spp, depth=depth, focal=35.0, fnumber=2.8,
             eye=(5.69, 1.4, 2.6), target=(0.0, 0.6, 0.0), up=(0.0, 1.0, 0.0), exposure=0.0,
             background=(0.0, 0.0, 0.0))
    return s, p


def write_params(path, p, threads=None, probe_pixels=None, **override):
    """Write an oracle/params.hpp parameter file."""
    q = dict(p); q.update(override)
    lines = []
    for key, val in q.items():
        if key in ("max_batch_paths", "pool_paths") or key.startswith("_"):   # this library's own knobs (YartRenderParams) / test notes
            continue
        if isinstance(val, (tuple, list, np.ndarray)):
            lines.append(f"{key} " + " ".join(repr(float(v)) if key != "size" else str(int(v)) for v in val))
        elif isinstance(val, float):
            lines.append(f"{key} {val!r}")
        else:
            lines.append(f"{key} {int(val)}")
    if threads is not None:
        lines.append(f"threads {int(threads)}")
    if probe_pixels is not None:
        lines.append("probe_pixels " + " ".join(str(int(v)) for xy in probe_pixels for v in xy))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def instances(width=96, height=96, spp=4, depth=4, n_instances=70, groups=4, seed=7, child_extent=3.5, group_extent=1.0,
              scale=(0.3, 0.9)):
    """Cornell room + `n_instances` instanced unit boxes under `groups` transformed group nodes
    (rotation + non-uniform scale + translation, nested two levels): exercises the scene-graph walk
    (skip links, nested transform chains, world-space node culling, per-ray node candidate masks —
    64 nodes is their capacity, so 1 + 1 + groups + n_instances is chosen around it by the tests)."""
    s, p = cornell(width, height, spp, depth)
    rng = np.random.RandomState(seed)
    mats = [s.add_material(Material(base=tuple(rng.uniform(0.2, 0.9, 3)), roughness=float(rng.uniform(0.2, 1.0)),
                                    metallic=float(rng.rand() > 0.7))) for _ in range(4)]
    b = MeshBuilder()
    b.box((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), mats[0])
    cube = s.add_mesh(b.build())

    def trs(t, ry, sc):
        cs, sn = math.cos(ry), math.sin(ry)
        m = np.array([[cs * sc[0], 0, sn * sc[2], t[0]], [0, sc[1], 0, t[1]], [-sn * sc[0], 0, cs * sc[2], t[2]],
                      [0, 0, 0, 1]], np.float64)
        return m.astype(np.float32), np.linalg.inv(m).astype(np.float32)
    group_nodes = []
    for g in range(groups):
        f, i = trs((rng.uniform(-group_extent, group_extent), rng.uniform(0.5, 2.0) if group_extent == 1.0 else rng.uniform(0.5, 5.5),
                    rng.uniform(-group_extent, group_extent)), rng.uniform(-0.5, 0.5), (1.0, 1.0, 1.0))
        group_nodes.append((g, f, i))
    per = [n_instances // groups + (1 if g < n_instances % groups else 0) for g in range(groups)]
    for g, f, i in group_nodes:                       # pre-order: group, then its instances
        gi = s.add_node(-1, 0, f, i)
        for _ in range(per[g]):
            f2, i2 = trs((rng.uniform(-child_extent, child_extent), rng.uniform(0.0, 6.0) if child_extent == 3.5 else rng.uniform(-child_extent, child_extent),
                          rng.uniform(-child_extent, child_extent)), rng.uniform(0, 6.28), tuple(rng.uniform(scale[0], scale[1], 3)))
            s.add_node(cube, gi, f2, i2)
    s.create_area_lights()
    return s, p


def alpha_instances(width=96, height=96, spp=4, depth=5, n_instances=9, cards=220, seed=11, tex=32):
    """Cornell room + instances of a "bush" (`cards` small alpha cut-out quads at random in a unit cube: a tree of its own
    ten levels deep, every leaf an alpha candidate) and of a thin-glass pane, each under a rotated, non-uniformly scaled node
    of a translated group: rays meet alpha-tested and NEE-transparent candidates inside TRANSFORMED nodes, with and without a
    hit found before, with traversal stacks of some depth — the cases the hand-over from the lean to the general traversal
    kernels (resume records, trace_lean.hpp) has to carry; the opaque walls behind give closest-hit rays their earlier hits."""
    s, p = cornell(width, height, spp, depth)
    rng = np.random.RandomState(seed)
    leaf_tex = s.add_texture(tex_base_color(tex, 5, (0.1, 0.35, 0.08), (0.4, 0.6, 0.2), alpha=tex_alpha_leaves(tex, 6)))
    leaf = s.add_material(Material(base=(1, 1, 1), roughness=0.8, tex_base=leaf_tex))
    glass = s.add_material(Material(base=(0.8, 0.9, 1.0), transmission=1.0, roughness=0.0, ior=1.5, thin_transmission=True))
    solid = s.add_material(Material(base=(0.7, 0.3, 0.2), roughness=0.6))
    b = MeshBuilder()
    for _ in range(cards):
        c = rng.uniform(-0.5, 0.5, 3)
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        v = np.cross(u, rng.normal(size=3)); v /= np.linalg.norm(v)
        h = rng.uniform(0.06, 0.16)
        b.quad(c - h * u - h * v, c + h * u - h * v, c + h * u + h * v, c - h * u + h * v, leaf, uv_scale=1.0)
    b.box((-0.12, -0.5, -0.12), (0.12, 0.1, 0.12), solid)          # a trunk: opaque hits inside the same tree
    bush = s.add_mesh(b.build())
    g = MeshBuilder()
    g.quad((-0.5, -0.5, 0.0), (0.5, -0.5, 0.0), (0.5, 0.5, 0.0), (-0.5, 0.5, 0.0), glass)
    pane = s.add_mesh(g.build())

    def xf(t, ry, rx, sc):
        cy, sy, cx, sx = math.cos(ry), math.sin(ry), math.cos(rx), math.sin(rx)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        m = np.eye(4); m[:3, :3] = Ry @ Rx @ np.diag(sc); m[:3, 3] = t
        return m.astype(np.float32), np.linalg.inv(m).astype(np.float32)
    f, i = xf((0.3, 0.0, -0.4), 0.35, 0.0, (1.0, 1.0, 1.0))
    group = s.add_node(-1, 0, f, i)
    for k in range(n_instances):
        t = (rng.uniform(-3.0, 3.0), rng.uniform(0.8, 4.5), rng.uniform(-3.0, 2.5))
        f2, i2 = xf(t, rng.uniform(0, 6.28), rng.uniform(-0.6, 0.6), tuple(rng.uniform(0.9, 2.2, 3)))
        s.add_node(pane if k % 4 == 3 else bush, group, f2, i2)
    s.create_area_lights()
    return s, p


def many_records(width=96, height=96, spp=4, depth=5, n_materials=80, n_lights=12, seed=3, tex=32):
    """Cornell room + a wall of `n_materials` small tiles, each with a material and a base-colour texture of its own (metal / rough /
    clearcoat mixes), and `n_lights` small emissive quads (two area lights each): more materials (> 64), texture descriptors (> 128
    with the walls') and lights (> 8) than the slots of the LDS copies the shade kernel keeps of these tables (wavefront_kernels.inc)
    — the tables then stay in memory, and the frame must not change."""
    s, p = cornell(width, height, spp, depth)
    rng = np.random.RandomState(seed)
    b = MeshBuilder()
    cols = 10
    for k in range(n_materials):
        t = s.add_texture(tex_base_color(tex, 100 + k, tuple(rng.uniform(0.1, 0.5, 3)), tuple(rng.uniform(0.5, 0.95, 3))))
        t2 = s.add_texture(tex_metal_rough(tex, 300 + k, (0.2, 0.9), (0.0, 1.0)))
        m = s.add_material(Material(base=(1, 1, 1), roughness=float(rng.uniform(0.15, 1.0)), metallic=float(rng.rand() > 0.6),
                                    clearcoat=float(rng.rand() > 0.8), tex_base=t, tex_mr=t2 if k % 3 == 0 else -1))
        x0 = -4.5 + 0.9 * (k % cols); y0 = 0.6 + 0.9 * (k // cols)
        b.quad((x0, y0, -4.6), (x0 + 0.8, y0, -4.6), (x0 + 0.8, y0 + 0.8, -4.55), (x0, y0 + 0.8, -4.55), m, uv_scale=1.0)
    for k in range(n_lights):
        e = s.add_material(Material(base=(0.8, 0.8, 0.8), roughness=1.0, emission=tuple(rng.uniform(2.0, 9.0, 3))))
        x0 = -4.2 + 0.7 * k
        b.quad((x0, 9.9, 1.0), (x0 + 0.4, 9.9, 1.0), (x0 + 0.4, 9.9, 1.5), (x0, 9.9, 1.5), e)
    s.add_node(s.add_mesh(b.build()))
    s.create_area_lights()
    return s, p


def stacked_leaves(width=96, height=96, spp=4, depth=4, stacks=(40, 32, 64, 31, 33)):
    """Cornell room + cards made of `n` coincident copies of the same two triangles (n from `stacks`), the
    copies cycling through four materials. Every copy of a triangle has the same centroid, so the reference's
    SAH split degenerates and the node stays a leaf of arbitrary span (bvh.hpp:159-161) — beyond the 5-bit span
    field of the device's traversal-stack link word (traverse.hpp::packLink / leafSpan); of several equal-t hits
    the first in leaf order wins (`hit.t <= t` rejects the later ones, ray-integrator.cpp:196)."""
    s, p = cornell(width, height, spp, depth)
    mats = [s.add_material(Material(base=c, roughness=r, metallic=m)) for c, r, m in
            (((0.9, 0.2, 0.2), 1.0, 0.0), ((0.2, 0.9, 0.2), 0.4, 0.0), ((0.2, 0.3, 0.9), 0.2, 1.0), ((0.9, 0.8, 0.2), 0.7, 0.0))]
    b = MeshBuilder()
    for k, n in enumerate(stacks):
        x0 = -4.0 + 8.0 * k / max(1, len(stacks)); x1 = x0 + 6.0 / max(1, len(stacks))
        y0, y1, z = 1.0 + 0.9 * (k % 3), 3.5 + 0.9 * (k % 3), 1.0 - 0.7 * k
        for c in range(n):
            b.quad((x0, y0, z), (x1, y0, z), (x1, y1, z), (x0, y1, z), mats[(c + k) % 4])
    s.add_node(s.add_mesh(b.build()))
    s.create_area_lights()
    return s, p


def random_scene(seed, width=64, height=48, spp=4, depth=6, crowd=0, extras=False):
    """Seeded random scene of the parity fuzz (tests/test_fuzz_scenes.py): every constructor argument of the reference's
    ParametricBSDF (bsdf/parametric.hpp:16-37) drawn at random — with the end points 0 and 1 over-represented, they pick other
    branches of parametric.cpp —, every texture kind at odd, non-square sizes (texture.hpp:105-161), alpha cut-outs, thin and
    refractive transmission with volume absorption, emissive triangles with and without emission textures, a triangle soup next
    to smooth shapes, instances under nested non-uniformly scaled and rotated nodes (ray-integrator.cpp:20-54), zero to two
    infinite lights, pinhole and thin-lens cameras (camera.hpp:138-164). No two seeds share a code path mix; nothing is tuned
    to look good. crowd > 0: that many further instance nodes of small meshes under nested groups (64 nodes and more: the lean
    kernels' other scene-graph walks — top-level hierarchy, chunked candidate masks, per-lane node walk). extras: geometry that
    makes ORDER matter — coincident duplicates with different materials (equal-t candidates, also cut-outs on top of opaque
    twins), zero-area and needle triangles, planes at exactly 0 — and instances at extreme scales (1e-3 .. 1e3: the padded world
    boxes and intervals of the conservative culls)."""
    rng = np.random.RandomState(seed)
    s = Scene()
    U = rng.uniform

    def pick(*vals):
        return vals[rng.randint(len(vals))]

    def rtex(channels, typ, lo=0, hi=255):
        h, w = int(rng.randint(2, 41)), int(rng.randint(2, 41))
        if rng.rand() < 0.5:          # smooth: a few random texels blown up
            g = rng.randint(lo, hi + 1, (int(rng.randint(1, 5)), int(rng.randint(1, 5)), channels))
            yy = (np.arange(h) * g.shape[0] // h)[:, None]; xx = (np.arange(w) * g.shape[1] // w)[None, :]
            data = g[yy, xx]
        else:
            data = rng.randint(lo, hi + 1, (h, w, channels))
        return data.astype(np.uint8), typ

    def add(data_typ):
        return s.add_texture(Texture(*data_typ))

    materials = []
    n_mat = int(rng.randint(3, 11))
    for k in range(n_mat):
        m = Material(base=tuple(U(0.0, 1.0, 3)), metallic=float(pick(0.0, 0.0, 1.0, U(0, 1))),
                     roughness=float(pick(0.0, 1.0, U(0, 1), U(0, 0.3))), transmission=float(pick(0.0, 0.0, 0.0, 1.0, U(0, 1))),
                     ior=float(pick(1.5, 1.33, U(1.05, 2.4))), anisotropic=float(pick(0.0, 0.0, U(0, 1))),
                     aniso_rotation=float(U(0, 1)), clearcoat=float(pick(0.0, 0.0, 1.0, U(0, 1))),
                     clearcoat_roughness=float(pick(0.0, U(0, 0.6))), normal_scale=float(pick(1.0, U(0.2, 1.6))),
                     thin_transmission=bool(rng.rand() < 0.4), volume_color=tuple(U(0.2, 1.0, 3)),
                     volume_density=float(pick(0.0, U(0, 2.0))))
        if rng.rand() < 0.2:
            m.emission = tuple(U(0.5, 12.0, 3))
        if rng.rand() < 0.4:
            data, typ = rtex(4, TEX_SRGB)
            if rng.rand() < 0.55:
                data[..., 3] = 255                      # opaque: no alpha test
            elif rng.rand() < 0.5:
                data[..., 3] = np.where(data[..., 3] < 128, 0, 255)
            m.tex_base = add((data, typ))
        if rng.rand() < 0.35: m.tex_mr = add(rtex(2, TEX_NONCOLOR))
        if rng.rand() < 0.3: m.tex_transmission = add(rtex(1, TEX_NONCOLOR))
        if rng.rand() < 0.35: m.tex_normal = add(rtex(3, TEX_NONCOLOR, 64, 255))
        if rng.rand() < 0.3: m.tex_clearcoat = add(rtex(1, TEX_NONCOLOR))
        if m.is_emissive and rng.rand() < 0.5: m.tex_emission = add(rtex(3, TEX_SRGB))
        materials.append(s.add_material(m))
    M = lambda: materials[rng.randint(n_mat)]

    def shapes(b, count, extent):
        for _ in range(count):
            c = U(-extent, extent, 3); c[1] = abs(c[1]) * 0.6 + 0.2
            kind = rng.randint(4)
            if kind == 0:
                b.sphere(tuple(c), float(U(0.3, 1.2)), M(), nu=int(rng.randint(5, 14)), nv=int(rng.randint(3, 8)),
                         uv_scale=(float(U(0.5, 3)), float(U(0.5, 3))))
            elif kind == 1:
                h = U(0.2, 1.0, 3)
                b.box(tuple(c - h), tuple(c + h), M(), rot_y=float(U(0, 3.1)))
            elif kind == 2:       # an upright card (cut-outs, thin glass)
                a = U(0, 6.28); r = U(0.5, 1.6); t = np.array([math.cos(a), 0.0, math.sin(a)]) * r
                up = np.array([0.0, U(0.6, 2.0), 0.0])
                b.quad(tuple(c - t), tuple(c + t), tuple(c + t + up), tuple(c - t + up), M(), uv_scale=float(U(0.5, 3)))
            else:                 # a triangle soup: random, partly degenerate-thin triangles with random shading normals
                k = int(rng.randint(3, 40))
                p = c + U(-1.0, 1.0, (k, 3, 3)) * U(0.05, 1.0, (k, 1, 1))
                n = U(-1, 1, (k * 3, 3)); n /= np.linalg.norm(n, axis=-1, keepdims=True)
                tg = U(-1, 1, (k * 3, 4)); tg[:, 3] = np.where(tg[:, 3] < 0, -1.0, 1.0)
                b.add(p.reshape(-1, 3), n, tg, U(-2, 3, (k * 3, 2)), np.arange(k * 3).reshape(k, 3), M())

    # the main mesh: a floor (sometimes a closed room) + shapes
    b = MeshBuilder()
    L = float(U(4.0, 9.0))
    fm = M()
    b.grid(lambda A, B: ((A - 0.5) * 2 * L, 0 * A, (B - 0.5) * 2 * L), int(rng.randint(1, 7)), int(rng.randint(1, 7)), fm,
           uv_scale=(float(U(1, 5)), float(U(1, 5))), flip=True)
    if rng.rand() < 0.4:
        H = float(U(4.0, 8.0)); wm = M()
        b.quad((-L, 0, -L), (L, 0, -L), (L, H, -L), (-L, H, -L), wm)
        b.quad((-L, 0, L), (-L, 0, -L), (-L, H, -L), (-L, H, L), M())
        b.quad((L, 0, -L), (L, 0, L), (L, H, L), (L, H, -L), M())
        if rng.rand() < 0.5: b.quad((-L, H, -L), (L, H, -L), (L, H, L), (-L, H, L), wm)
    shapes(b, int(rng.randint(2, 9)), L * 0.6)
    s.add_node(s.add_mesh(b.build()), 0, *trs(tuple(U(-0.5, 0.5, 3)), tuple(U(-1, 1, 3) + 1e-3), float(pick(0.0, U(-0.3, 0.3))),
                                               pick((1, 1, 1), tuple(U(0.7, 1.4, 3)))))
    # instanced meshes under nested nodes
    for _ in range(int(rng.randint(0, 4))):
        bb = MeshBuilder()
        shapes(bb, int(rng.randint(1, 4)), 1.0)
        mesh = s.add_mesh(bb.build())
        parent = 0
        for _level in range(int(rng.randint(0, 3))):
            parent = s.add_node(-1, parent, *trs(tuple(U(-2, 2, 3)), tuple(U(-1, 1, 3) + 1e-3), float(U(-3.1, 3.1)),
                                                 tuple(U(0.5, 1.6, 3))))
        for _inst in range(int(rng.randint(1, 4))):
            t = U(-L * 0.5, L * 0.5, 3); t[1] = abs(t[1]) * 0.3
            s.add_node(mesh, parent, *trs(tuple(t), tuple(U(-1, 1, 3) + 1e-3), float(U(-3.1, 3.1)),
                                          pick(tuple(U(0.4, 1.8, 3)), (1, 1, 1), (float(U(0.3, 2)),) * 3)))
    if extras:
        be = MeshBuilder()
        for _ in range(int(rng.randint(1, 5))):        # coincident twins / triplets: same vertices, other materials, rotated vertex order
            c = U(-2, 2, 3); c[1] = abs(c[1]) + 0.1
            a = U(0, 6.28); t = np.array([math.cos(a), 0.0, math.sin(a)]) * U(0.4, 1.5); up = np.array([0.0, U(0.5, 1.8), 0.0])
            q = [c - t, c + t, c + t + up, c - t + up]
            for k in range(int(rng.randint(2, 4))):
                r = int(rng.randint(4)) if k else 0
                be.quad(*[tuple(q[(i + r) % 4]) for i in range(4)], M(), uv_scale=float(U(0.5, 2)))
        # a slab whose faces lie at exactly x = 0 / y = 0 / z = 0, zero-area triangles, needles
        be.box((0.0, 0.0, 0.0), (float(U(0.3, 1.5)), float(U(0.3, 1.5)), float(U(0.3, 1.5))), M())
        k = int(rng.randint(2, 8))
        pz = U(-1.5, 1.5, (k, 3, 3)); pz[:, 2] = pz[:, 1]                                    # two equal vertices
        pn = U(-1.5, 1.5, (k, 3, 3)); pn[:, 2] = pn[:, 1] + U(-1e-6, 1e-6, (k, 3))           # needles
        pp_ = np.concatenate([pz, pn]).reshape(-1, 3)
        nn = U(-1, 1, (len(pp_), 3)); nn /= np.linalg.norm(nn, axis=-1, keepdims=True)
        be.add(pp_, nn, None, U(0, 1, (len(pp_), 2)), np.arange(len(pp_)).reshape(-1, 3), M())
        em = s.add_mesh(be.build())
        s.add_node(em, 0, *trs(tuple(U(-1, 1, 3) * (1, 0, 1)), (0, 1, 0), float(pick(0.0, U(-3, 3))), (1, 1, 1)))
        # the same mesh at extreme scales (their inverse transforms scale the ray the other way)
        for sc_ in (1e-3, 1e3, float(10 ** U(-2.5, 2.5))):
            g = s.add_node(-1, 0, *trs(tuple(U(-3, 3, 3) * (1, 0.1, 1)), tuple(U(-1, 1, 3) + 1e-3), float(U(-3.1, 3.1)), (sc_,) * 3))
            s.add_node(em, g, *trs(tuple(U(-1, 1, 3) * (1, 0, 1)), (0, 1, 0), 0.0, (1 / sc_,) * 3 if rng.rand() < 0.5 else (1, 1, 1)))
    if crowd:
        crowd_meshes = []
        for _ in range(int(rng.randint(1, 5))):
            bb = MeshBuilder()
            shapes(bb, int(rng.randint(1, 3)), 0.6)
            crowd_meshes.append(s.add_mesh(bb.build()))
        left = int(crowd)
        while left > 0:
            g = s.add_node(-1, 0, *trs(tuple(U(-L, L, 3) * (1, 0.2, 1)), tuple(U(-1, 1, 3) + 1e-3), float(U(-3.1, 3.1)),
                                       tuple(U(0.6, 1.5, 3))))
            left -= 1
            if rng.rand() < 0.3 and left > 0:
                g = s.add_node(-1, g, *trs(tuple(U(-1, 1, 3)), (0, 1, 0), float(U(-3.1, 3.1)), (1, 1, 1))); left -= 1
            for _inst in range(min(left, int(rng.randint(1, 12)))):
                t = U(-2.0, 2.0, 3); t[1] = abs(t[1]) * 0.5
                s.add_node(crowd_meshes[rng.randint(len(crowd_meshes))], g,
                           *trs(tuple(t), tuple(U(-1, 1, 3) + 1e-3), float(U(-3.1, 3.1)), pick((1, 1, 1), tuple(U(0.2, 0.9, 3)))))
                left -= 1
    s.create_area_lights()
    n_inf = 0
    if rng.rand() < 0.55:
        n = int(rng.randint(2, 25))
        sky = (U(0.0, 1.0, (n, n, 3)) ** 3 * float(U(0.5, 6.0))).astype(np.float32)
        if rng.rand() < 0.5: sky[rng.randint(n), rng.randint(n)] = U(20, 200, 3)       # a sun texel
        s.lights.append(Light(LIGHT_IMAGE_INF, texture=s.add_texture(Texture(sky, TEX_LINEAR)), radius=100.0)); n_inf += 1
    if rng.rand() < 0.3 or (n_inf == 0 and not s.lights):
        s.lights.append(Light(LIGHT_UNIFORM_INF, radius=100.0, emission=tuple(U(0.1, 1.5, 3))))
    a = U(0, 6.28); r = U(1.2, 2.2) * L; eye = (float(r * math.cos(a)), float(U(0.8, 5.0)), float(r * math.sin(a)))
    if rng.rand() < 0.25: eye = (float(U(-1, 1)), float(U(0.5, 2.0)), float(U(-1, 1)))    # inside the scene
    p = dict(size=(width, height), spp=spp, depth=depth, focal=float(pick(35.0, U(18, 85))),
             fnumber=float(pick(0.0, 0.0, U(1.2, 8.0))), aperture_sides=int(pick(0, 5, 6, 8)),
             eye=eye, target=(float(U(-1, 1)), float(U(0.3, 1.5)), float(U(-1, 1))), up=(0.0, 1.0, 0.0),
             exposure=float(pick(0.0, U(-1, 1))), background=tuple(float(v) for v in U(0, 0.3, 3)))
    return s, p


def fuzz_case(seed, width=64, height=48):
    """The parity fuzz's scene + settings for a seed (tests/test_fuzz_scenes.py, tools/fuzz_gpu.py): every third seed at 16 spp,
    every fourth at 12 bounces; some seeds with a crowd of instance nodes (70 / 260 / 1000 / 4300: the lean kernels' scene-graph
    walks for 64 nodes and more), every ninth with random_scene's `extras` (coincident duplicates, degenerate triangles, extreme scales)."""
    crowd = 4300 if seed % 211 == 210 else 1000 if seed % 101 == 100 else 260 if seed % 13 == 7 else 70 if seed % 7 == 4 else 0
    return random_scene(seed, width, height, 4 if seed % 3 else 16, 6 if seed % 4 else 12, crowd, extras=seed % 9 == 5)


def fuzz_frame_case(seed):
    """Second family of the parity fuzz: random_scene(seed) under random FRAME settings — sizes from 1x1 up, across the tile
    boundary, sample counts that are no powers of two (the sampler rounds log2(spp): sampler.hpp:84-100), progressive wave
    schedules (tile-renderer.hpp:121-124, 284-289), other tile sizes (the sampler's Morton index depends on it)."""
    rng = np.random.RandomState(seed ^ 0x5EED)
    pick = lambda *v: v[rng.randint(len(v))]
    w, h = int(rng.randint(1, 150)), int(rng.randint(1, 100))
    spp = int(pick(1, 2, 3, 4, 5, 7, 8, 12, 16, 24, 33, 64))
    s, p = random_scene(seed, w, h, spp, int(pick(1, 2, 3, 5, 8, 16)))
    p["first_wave"] = int(pick(spp, spp, 1, 2, 4)); p["max_wave"] = int(pick(spp, spp, 2, 4, 8, 16))
    p["tile"] = int(pick(64, 64, 64, 32, 16, 128))
    # device-side only (the reference has no such notion; write_params drops them): small batches, small path pools
    p["max_batch_paths"] = int(pick(0, 0, 257, 1000, 5000)); p["pool_paths"] = int(pick(0, 64, 640, 4096))
    return s, p
