"""Regenerates tests/golden/gltf/ — inputs for the glTF importer and the compiled reference's outputs for them.

Runs only where /root/reference is mounted (the build container):
    make -C oracle ref && python -c "import __graft_entry__ as g; g.build()" && python tests/golden/make_gltf_goldens.py

Inputs (written by tests/gltf_assets.py, all small):
    png_<case>.png          PNG files covering colour types 0/2/3/4/6, bit depths 1-16, every scanline filter, Adam7
                            interlace, tRNS keys and palette alpha, split IDAT
    jpg_<case>.jpg          baseline and progressive JPEG files from two encoders (stb_image_write via `yart_ref writejpg`, and
                            tests/gltf_assets.jpeg_encode): 4:4:4 / 4:2:0 / 4:2:2 / 4:4:0 / 4:1:1 / odd factors, grey, restart
                            intervals with fill bytes, one scan per component, RGB component ids, Adobe / JFIF signalling,
                            1- and 2-pixel-wide images, saturated chroma; progressive files (spectral selection, successive
                            approximation, EOB runs, restart intervals, partially refined)
    env_rle.hdr / env_flat.hdr / env_tiny.hdr   Radiance files (RLE scanlines, flat, width < 8)
    xform.txt               node T / R / S rows with parent links
    gallery.glb             a small scene using every material extension the importer maps, merged primitives, strided /
                            normalised / u8 / u16 accessors, generated indices, nested and matrix nodes, an instanced light mesh
    gallery.txt             camera / render parameters
Expected outputs (the reference's own code, through oracle/_ref/yart_ref):
    png_<case>.<C><type>.tex, jpg_<case>.<C><type>.tex    loadTexture<C>(file, type, channels) (core/texture.hpp:62-92): u32 w, h, C + bytes
    env_*.hdrtex                loadTextureHDR (core/texture.cpp:5-20): u32 w, h + float RGB
    xform.bin                   Transform(T*R*S) and node.transform * globalTransform per row (gltf.cpp:284-293)
    gallery.agx_golden.f32      gallery.f32 through the reference's AgX tonemapper, look golden (`yart_ref tonemap`)
    gallery.f32                 the reference's render of the scene the importer made of gallery.glb + env_rle.hdr
                                (the importer's .yscn is the input the reference gets: its loader cannot be built here)
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import gltf_assets as ga  # noqa: E402
from yart_amd import api, scenes  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "yart_ref")
OUT = os.path.join(HERE, "gltf")

# (C, type, channel list) combinations the importer asks for (gltf.cpp:62-176)
TEX_KINDS = {"4s": (4, 1, "0,1,2,3"), "2n": (2, 2, "1,2"), "1n": (1, 2, "0"), "3s": (3, 1, "0,1,2"), "3n": (3, 2, "0,1,2")}


def png_cases():
    rng = np.random.default_rng(11)
    w, h = 19, 13
    ramp = np.add.outer(np.arange(h) * 9, np.arange(w) * 13) % 256
    rgb = np.stack([ramp, (ramp * 3 + 40) % 256, rng.integers(0, 256, (h, w))], -1)
    alpha = (rng.integers(0, 4, (h, w)) * 85)[..., None]
    cases = {
        "rgba8": ga.png_encode(np.concatenate([rgb, alpha], -1), 6),
        "rgb8": ga.png_encode(rgb, 2, filters=(4, 3, 2, 1, 0)),
        "rgb8_key": ga.png_encode(np.where(rng.random((h, w, 1)) < 0.3, np.array([10, 200, 30]), rgb), 2,
                                  trns=bytes((0, 10, 0, 200, 0, 30))),
        "rgb8_adam7": ga.png_encode(rgb, 2, interlace=True, idat_split=97),
        "rgba16": ga.png_encode(rng.integers(0, 65536, (h, w, 4)), 6, depth=16),
        "rgb16_key": ga.png_encode(np.where(rng.random((h, w, 1)) < 0.3, np.array([300, 40000, 7]),
                                            rng.integers(0, 65536, (h, w, 3))), 2, depth=16,
                                   trns=bytes((1, 44, 156, 64, 0, 7))),
        "gray8": ga.png_encode(ramp[..., None], 0, filters=(3,)),
        "gray16": ga.png_encode(rng.integers(0, 65536, (h, w, 1)), 0, depth=16, filters=(4,)),
        "gray4_key": ga.png_encode(rng.integers(0, 16, (h, w, 1)), 0, depth=4, trns=bytes((0, 5))),
        "gray2": ga.png_encode(rng.integers(0, 4, (h, w, 1)), 0, depth=2, interlace=True),
        "gray1": ga.png_encode(rng.integers(0, 2, (h, w, 1)), 0, depth=1),
        "graya8": ga.png_encode(np.concatenate([ramp[..., None], alpha], -1), 4, filters=(1, 4)),
        "graya16": ga.png_encode(rng.integers(0, 65536, (h, w, 2)), 4, depth=16),
        "pal8": ga.png_encode(rng.integers(0, 40, (h, w, 1)), 3, palette=rng.integers(0, 256, (40, 3)),
                              trns=bytes(rng.integers(0, 256, 25).tolist())),
        "pal4": ga.png_encode(rng.integers(0, 16, (h, w, 1)), 3, depth=4, palette=rng.integers(0, 256, (16, 3))),
        "pal2_adam7": ga.png_encode(rng.integers(0, 4, (h, w, 1)), 3, depth=2, palette=rng.integers(0, 256, (4, 3)),
                                    trns=bytes((0, 128)), interlace=True),
        "pal1": ga.png_encode(rng.integers(0, 2, (5, 3, 1)), 3, depth=1, palette=[[255, 0, 0], [0, 0, 255]]),
    }
    return cases


def jpeg_cases(tmp_png):
    """Baseline JPEG files from two encoders: stb_image_write (through `yart_ref writejpg`: 4:2:0 and 4:4:4) and
    tests/gltf_assets.jpeg_encode (other samplings, restart intervals, several scans, colour-space signalling)."""
    rng = np.random.default_rng(21)
    h, w = 37, 53                                        # not multiples of the MCU size
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 70 * np.sin(xx / 5.0) + 40 * np.cos(yy / 3.0), 128 + 60 * np.sin((xx + yy) / 7.0),
                    128 + 80 * np.cos(xx / 4.0 - yy / 6.0)], -1) + rng.normal(0, 8, (h, w, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    with open(tmp_png, "wb") as f:
        f.write(ga.png_encode(img, 2))
    cases = {}
    for name, q in (("stb420", 75), ("stb444", 95), ("stb420_q20", 20)):
        out = tmp_png + ".jpg"
        subprocess.run([REF, "writejpg", tmp_png, str(q), out], check=True)
        cases[name] = open(out, "rb").read()
        os.remove(out)
    e = ga.jpeg_encode
    cases.update({
        "444": e(img),
        "420_rst": e(img, sampling=((2, 2), (1, 1), (1, 1)), restart=3, fill_bytes=True),
        "422": e(img, sampling=((2, 1), (1, 1), (1, 1)), quant=(3, 5)),
        "440": e(img, sampling=((1, 2), (1, 1), (1, 1)), restart=1),
        "411": e(img, sampling=((4, 1), (1, 1), (1, 1))),
        "chroma_v4": e(img, sampling=((1, 4), (1, 2), (1, 1))),
        "gray": e(img[..., :1]),
        "gray_h2v2": e(img[..., :1], sampling=((2, 2),)),
        "scans": e(img, sampling=((2, 2), (1, 1), (1, 1)), interleaved=False, restart=5),
        "rgb_ids": e(img, component_ids=(ord("R"), ord("G"), ord("B")), sampling=((2, 1), (1, 1), (2, 1))),
        "adobe0": e(img, jfif=False, adobe_transform=0),
        "adobe0_jfif": e(img, jfif=True, adobe_transform=0),
        "adobe1": e(img, jfif=False, adobe_transform=1, sampling=((2, 2), (1, 1), (1, 1))),
        "nojfif": e(img, jfif=False),
        "w1": e(img[:5, :1], sampling=((2, 2), (1, 1), (1, 1))),
        "w2": e(img[:3, :2], sampling=((2, 2), (1, 1), (1, 1))),
        "w2_422": e(img[:9, :2], sampling=((2, 1), (1, 1), (1, 1))),
        "sat": e(np.where((xx // 6 + yy // 5)[..., None] % 2 == 0, np.array([250, 10, 250]), np.array([5, 250, 8])).astype(np.uint8),
                 sampling=((2, 2), (1, 1), (1, 1)), quant=(2, 2)),      # saturated chroma: exercises the clamps
    })
    pe = ga.jpeg_encode_progressive
    cases.update({
        "prog_444": pe(img),
        "prog_420": pe(img, sampling=((2, 2), (1, 1), (1, 1)), quant=(3, 4)),
        "prog_420_rst": pe(img, sampling=((2, 2), (1, 1), (1, 1)), restart=4),
        "prog_422_partial": pe(img, sampling=((2, 1), (1, 1), (1, 1)),      # stops before the last refinement passes
                               script=[("dc", 0, 2), ("ac", 0, 1, 9, 0, 1), ("ac", 1, 1, 63, 0, 2), ("ac", 2, 1, 20, 0, 0),
                                       ("dc", 2, 1), ("ac", 0, 10, 63, 0, 3), ("ac", 0, 10, 63, 3, 2)]),
        "prog_gray": pe(img[..., :1]),
        "prog_w2": pe(img[:11, :2], sampling=((2, 2), (1, 1), (1, 1))),
        "prog_flat": pe(np.full((40, 40, 3), 77, np.uint8), sampling=((2, 2), (1, 1), (1, 1))),   # long EOB runs
    })
    os.remove(tmp_png)
    return cases


def sky(size):
    """Small octahedral environment: gradient + a bright lobe (values spanning many RGBE exponents)."""
    v, u = np.meshgrid((np.arange(size) + 0.5) / size, (np.arange(size) + 0.5) / size, indexing="ij")
    base = np.stack([0.3 + 0.5 * u, 0.4 + 0.3 * v, 0.9 - 0.4 * u * v], -1)
    lobe = np.exp(-((u - 0.7) ** 2 + (v - 0.3) ** 2) * 60.0)[..., None] * np.array([90.0, 70.0, 40.0])
    return (base + lobe).astype(np.float32)


def build_gallery(path):
    """The test scene, in glTF terms. Returns the camera / render parameters."""
    rng = np.random.default_rng(5)
    b = ga.GltfBuilder()
    # --- textures --------------------------------------------------------------------------------
    n = 32
    yy, xx = np.mgrid[0:n, 0:n]
    checker = ((xx // 4 + yy // 4) % 2)[..., None]
    base_rgb = np.where(checker == 1, np.array([200, 120, 60]), np.array([230, 220, 190])) + rng.integers(-12, 12, (n, n, 3))
    base = np.concatenate([np.clip(base_rgb, 0, 255), np.full((n, n, 1), 255)], -1)
    leaf_alpha = (((xx - 16) ** 2 + (yy - 16) ** 2) < 150).astype(np.int64)[..., None] * 255
    leaf = np.concatenate([np.stack([40 + xx, 150 + yy * 2, 30 + xx // 2], -1), leaf_alpha], -1)
    mr = np.stack([np.zeros((16, 16), np.int64), rng.integers(60, 230, (16, 16)), rng.integers(0, 2, (16, 16)) * 255], -1)
    nrm = np.stack([128 + rng.integers(-30, 30, (16, 16)), 128 + rng.integers(-30, 30, (16, 16)), np.full((16, 16), 235)], -1)
    emis = np.stack([np.full((8, 8), 255), 180 + rng.integers(0, 60, (8, 8)), 90 + rng.integers(0, 60, (8, 8))], -1)
    trans = rng.integers(120, 256, (8, 8, 1))
    t_base = b.texture(b.image(ga.png_encode(base, 6)))
    t_leaf = b.texture(b.image(ga.png_encode(leaf, 6, interlace=True)))
    t_mr = b.texture(b.image(ga.png_encode(mr, 2, filters=(4,))))
    t_nrm = b.texture(b.image(ga.png_encode(nrm, 2)))
    t_em = b.texture(b.image(ga.png_encode(emis, 2, filters=(2, 1))))
    t_tr = b.texture(b.image(ga.png_encode(trans, 0)))
    t_uri = b.texture(b.image(None, uri="missing.png"))           # image by URI: the reference ends up without a texture
    # --- materials -------------------------------------------------------------------------------
    m_floor = b.material(name="floor", pbrMetallicRoughness={"baseColorTexture": {"index": t_base}, "roughnessFactor": 0.9,
                                                              "metallicFactor": 0.2, "metallicRoughnessTexture": {"index": t_mr}},
                         normalTexture={"index": t_nrm, "scale": 0.7})
    m_light = b.material(name="lamp", pbrMetallicRoughness={"baseColorFactor": [0.8, 0.8, 0.8, 1.0], "metallicFactor": 0.0},
                         emissiveFactor=[1.0, 0.8, 0.6], emissiveTexture={"index": t_em},
                         extensions={"KHR_materials_emissive_strength": {"emissiveStrength": 14.0}})
    m_glass = b.material(name="glass", pbrMetallicRoughness={"baseColorFactor": [0.95, 0.98, 1.0, 1.0], "roughnessFactor": 0.05,
                                                              "metallicFactor": 0.0},
                         extensions={"KHR_materials_transmission": {"transmissionFactor": 0.95, "transmissionTexture": {"index": t_tr}},
                                     "KHR_materials_ior": {"ior": 1.45},
                                     "KHR_materials_volume": {"attenuationColor": [0.7, 0.9, 0.8], "attenuationDistance": 0.6,
                                                              "thicknessFactor": 1.0}})
    m_coat = b.material(name="coated", pbrMetallicRoughness={"baseColorFactor": [0.7, 0.1, 0.08, 1.0], "roughnessFactor": 0.35,
                                                              "metallicFactor": 1.0, "baseColorTexture": {"index": t_uri}},
                        extensions={"KHR_materials_clearcoat": {"clearcoatFactor": 0.8, "clearcoatRoughnessFactor": 0.1},
                                    "KHR_materials_anisotropy": {"anisotropyStrength": 0.6, "anisotropyRotation": 0.4}})
    m_leaf = b.material(name="leaf", pbrMetallicRoughness={"baseColorTexture": {"index": t_leaf}, "roughnessFactor": 0.8,
                                                            "metallicFactor": 0.0}, alphaMode="MASK", doubleSided=True)
    m_plain = b.material(name="defaults")                                    # every factor at its glTF default

    def quad(p0, p1, p2, p3, uvs=1.0):
        p = np.array([p0, p1, p2, p3], np.float32)
        nrmv = np.cross(p[1] - p[0], p[3] - p[0]); nrmv = (nrmv / np.linalg.norm(nrmv)).astype(np.float32)
        tg = (p[1] - p[0]) / np.linalg.norm(p[1] - p[0])
        return (p, np.tile(nrmv, (4, 1)), np.tile(np.append(tg, 1.0).astype(np.float32), (4, 1)),
                (np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * uvs), np.array([0, 1, 2, 0, 2, 3], np.uint32))

    def prim(p, nv, tg, uv, idx, material, tangents=True, idx_dtype=np.uint32, uv_norm=False, stride=None, indexed=True, mode=None):
        at = {"POSITION": b.accessor(p, "VEC3", min_max=True, stride=stride, byte_offset=8 if stride else 0),
              "NORMAL": b.accessor(nv, "VEC3")}
        if uv_norm:
            at["TEXCOORD_0"] = b.accessor(np.round(np.clip(uv, 0, 1) * 65535).astype(np.uint16), "VEC2", normalized=True, stride=8)
        else:
            at["TEXCOORD_0"] = b.accessor(uv.astype(np.float32), "VEC2")
        if tangents:
            at["TANGENT"] = b.accessor(tg, "VEC4")
        d = {"attributes": at}
        if material is not None:
            d["material"] = material
        if indexed:
            d["indices"] = b.accessor(idx.astype(idx_dtype), "SCALAR")
        if mode is not None:
            d["mode"] = mode
        return d

    # room: floor + back wall (two primitives, one of them strided) + a LINES primitive the loader skips
    floor = quad((-4, 0, 4), (4, 0, 4), (4, 0, -4), (-4, 0, -4), uvs=3.0)
    wall = quad((-4, 0, -4), (4, 0, -4), (4, 5, -4), (-4, 5, -4))
    lines = prim(*quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)), material=m_plain, mode=1)
    mesh_room = b.mesh([prim(*floor, material=m_floor), lines, prim(*wall, material=m_coat, stride=20, idx_dtype=np.uint16)])
    # lamp: one quad, u8 indices
    lamp = quad((-0.6, 0, -0.6), (-0.6, 0, 0.6), (0.6, 0, 0.6), (0.6, 0, -0.6))
    mesh_lamp = b.mesh([prim(*lamp, material=m_light, tangents=False, idx_dtype=np.uint8)])
    # sphere: u16 indices, normalised u16 texcoords
    nu, nvv = 24, 12
    th, ph = np.meshgrid(np.linspace(0, 2 * np.pi, nu + 1), np.linspace(0, np.pi, nvv + 1), indexing="xy")
    sp = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], -1).reshape(-1, 3).astype(np.float32)
    suv = np.stack([th / (2 * np.pi), ph / np.pi], -1).reshape(-1, 2)
    stg = np.concatenate([np.stack([-np.sin(th), 0 * th, np.cos(th)], -1).reshape(-1, 3), np.ones((sp.shape[0], 1))], -1).astype(np.float32)
    sidx = []
    for j in range(nvv):
        for i in range(nu):
            a = j * (nu + 1) + i
            sidx += [a, a + nu + 1, a + 1, a + 1, a + nu + 1, a + nu + 2]
    mesh_sphere = b.mesh([prim(sp, sp.copy(), stg, suv, np.array(sidx), material=m_glass, idx_dtype=np.uint16, uv_norm=True)])
    # leaf card: non-indexed triangles (indices are generated), default material index (primitive without `material`)
    lq = quad((-1, 0, 0), (1, 0, 0), (1, 2, 0), (-1, 2, 0))
    order = lq[4]
    mesh_leaf = b.mesh([prim(lq[0][order], lq[1][order], lq[2][order], lq[3][order], None, material=m_leaf, indexed=False)])
    mesh_default = b.mesh([prim(*quad((-1, 0, 1), (1, 0, 1), (1, 0.0, -1), (-1, 0.0, -1)), material=None)])   # value_or(0)
    # --- nodes -----------------------------------------------------------------------------------
    n_leaf = b.node(mesh_leaf, translation=(0.3, 0.0, 0.9), rotation=ga.quat_axis_angle((0, 1, 0), 0.5), scale=(0.5, 0.6, 0.5))
    n_sphere = b.node(mesh_sphere, translation=(1.2, 0.8, 0.4), scale=(0.8, 0.8, 0.8), children=[n_leaf])
    n_lamp = b.node(mesh_lamp, translation=(0.0, 4.2, 0.0), rotation=ga.quat_axis_angle((1, 0, 0.2), np.pi - 0.15))
    b.node(mesh_room, translation=(0.0, 0.0, -0.5), rotation=ga.quat_axis_angle((0, 1, 0), 0.2), children=[n_lamp, n_sphere], root=True)
    # a second lamp instance through a matrix node (column-major: scale 0.5 / 1 / 0.5, then translate) under a flipped parent
    n_lamp2 = b.node(mesh_lamp, matrix=[0.5, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0.5, 0, -2.0, -3.0, 1.0, 1])
    b.node(None, rotation=ga.quat_axis_angle((0, 0, 1), np.pi), children=[n_lamp2], root=True)
    b.node(mesh_default, translation=(-2.2, 0.02, 1.5), root=True)
    b.write_glb(path)
    return dict(size=(96, 72), spp=16, depth=6, focal=30.0, fnumber=0.0, eye=(0.5, 2.2, 7.5), target=(0.3, 1.4, 0.0),
                up=(0.0, 1.0, 0.0), exposure=0.0, background=(0.0, 0.0, 0.0))


def main():
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/yart_ref missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    for name, data in png_cases().items():
        path = os.path.join(OUT, f"png_{name}.png")
        with open(path, "wb") as f:
            f.write(data)
        for tag, (c, typ, ch) in TEX_KINDS.items():
            if tag != "4s" and name not in ("rgba8", "rgb16_key", "pal8", "graya8"):
                continue                   # the channel-selection variants on a few files only
            subprocess.run([REF, "texture", path, str(c), str(typ), ch, os.path.join(OUT, f"png_{name}.{tag}.tex")], check=True)
    for name, data in jpeg_cases(os.path.join(OUT, "_src.png")).items():
        path = os.path.join(OUT, f"jpg_{name}.jpg")
        with open(path, "wb") as f:
            f.write(data)
        for tag in ("4s",) + (("3n", "2n") if name in ("stb420", "422") else ()):
            c, typ, ch = TEX_KINDS[tag]
            subprocess.run([REF, "texture", path, str(c), str(typ), ch, os.path.join(OUT, f"jpg_{name}.{tag}.tex")], check=True)
    rgbe = ga.rgbe_from_float(sky(16))
    hdrs = {"env_rle": ga.hdr_encode(rgbe, rle=True), "env_flat": ga.hdr_encode(rgbe, rle=False, magic=b"#?RGBE"),
            "env_tiny": ga.hdr_encode(ga.rgbe_from_float(sky(6)), rle=False)}
    for name, data in hdrs.items():
        with open(os.path.join(OUT, name + ".hdr"), "wb") as f:
            f.write(data)
        subprocess.run([REF, "hdr", os.path.join(OUT, name + ".hdr"), os.path.join(OUT, name + ".hdrtex")], check=True)
    rng = np.random.default_rng(3)
    rows = []
    for i in range(24):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        t = rng.normal(size=3) * 3
        s = np.exp(rng.normal(size=3) * 0.5) * (1 if i % 5 else -1)
        if i == 0:
            t, q, s = np.zeros(3), np.array([0, 0, 0, 1.0]), np.ones(3)
        rows.append(" ".join(repr(float(np.float32(v))) for v in (*t, *q, *s)) + f" {(i - 1) // 2 if i else -1}")
    with open(os.path.join(OUT, "xform.txt"), "w") as f:
        f.write("\n".join(rows) + "\n")
    subprocess.run([REF, "xform", os.path.join(OUT, "xform.txt"), os.path.join(OUT, "xform.bin")], check=True)
    p = build_gallery(os.path.join(OUT, "gallery.glb"))
    scenes.write_params(os.path.join(OUT, "gallery.txt"), p, threads=8)
    tmp = os.path.join(OUT, "_gallery.yscn")
    api.gltf_to_yscn(os.path.join(OUT, "gallery.glb"), tmp, env_hdr=os.path.join(OUT, "env_rle.hdr"), env_radius=100.0)
    subprocess.run([REF, "render", tmp, os.path.join(OUT, "gallery.txt"), os.path.join(OUT, "gallery.f32")], check=True,
                   stdout=subprocess.DEVNULL)
    os.remove(tmp)
    # the reference's own AgX (host) on that frame: what a renderer with `tonemapper = &agx` leaves in its buffer
    subprocess.run([REF, "tonemap", os.path.join(OUT, "gallery.f32"), str(p["size"][0]), str(p["size"][1]), "golden",
                    os.path.join(OUT, "gallery.agx_golden.f32"), os.devnull], check=True)
    print({f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
