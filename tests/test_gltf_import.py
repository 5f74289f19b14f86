"""glTF 2.0 / GLB importer (reference src/gltf/gltf.cpp:319-358 + core/texture.hpp:62-92).

The reference's loader itself cannot be built here (fastgltf is not vendored), so the importer is pinned piecewise
against the parts of the reference that can: its texture load (stb_image decode + gamma-2 re-encode), its .hdr load,
its float4x4 / Transform algebra — goldens under tests/golden/gltf/ made by oracle/_ref — and, on the GPU, the
reference's render of the imported scene. The mapping asset -> scene (gltf.cpp:62-317) is checked against the
source's behaviour case by case.
"""
import os
import struct

import numpy as np
import pytest

from tests import gltf_assets as ga
from tests.conftest import GOLDEN
from tests.paramfile import load_params

G = os.path.join(GOLDEN, "gltf")


@pytest.fixture(scope="module")
def api(built):
    from yart_amd import api
    return api


def load(api, path, tmp_path, **kw):
    from yart_amd import yscn
    out = os.path.join(tmp_path, "out.yscn")
    api.gltf_to_yscn(path, out, **kw)
    return yscn.Scene.load(out)


# slot of the material that makes the importer ask for a (C, type, channels) combination (gltf.cpp:62-146)
def _material_for(kind, tex):
    if kind == "4s": return dict(pbrMetallicRoughness={"baseColorTexture": {"index": tex}})
    if kind == "2n": return dict(pbrMetallicRoughness={"metallicRoughnessTexture": {"index": tex}})
    if kind == "1n": return dict(extensions={"KHR_materials_transmission": {"transmissionTexture": {"index": tex}}})
    if kind == "3s": return dict(emissiveTexture={"index": tex})
    return dict(normalTexture={"index": tex})


def _triangle_mesh(b, material):
    p = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    n = np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    return b.mesh([{"attributes": {"POSITION": b.accessor(p, "VEC3"), "NORMAL": b.accessor(n, "VEC3"),
                                   "TEXCOORD_0": b.accessor(p[:, :2].copy(), "VEC2")}, "material": material}])


PNG_TEX = sorted(f[:-4] for f in os.listdir(G) if f.endswith(".tex")) if os.path.isdir(G) else []


@pytest.mark.parametrize("case", PNG_TEX)
def test_image_texture_equals_reference_loadtexture(api, tmp_path, case):
    """Embedded PNG / baseline JPEG -> texture bytes == the reference's loadTexture<C> on the same file
    (stb_image decode + gamma-2 re-encode), byte for byte."""
    name, kind = case.rsplit(".", 1)
    with open(os.path.join(G, name + (".png" if name.startswith("png_") else ".jpg")), "rb") as f:
        png = f.read()
    b = ga.GltfBuilder()
    mat = b.material(**_material_for(kind, b.texture(b.image(png))))
    b.node(_triangle_mesh(b, mat), root=True)
    glb = os.path.join(tmp_path, "t.glb")
    b.write_glb(glb)
    s = load(api, glb, tmp_path)
    with open(os.path.join(G, case + ".tex"), "rb") as f:
        raw = f.read()
    w, h, c = struct.unpack_from("<3I", raw)
    want = np.frombuffer(raw, np.uint8, offset=12).reshape(h, w, c)
    assert len(s.textures) == 1
    t = s.textures[0]
    assert t.data.shape == (h, w, c) and t.data.dtype == np.uint8
    assert t.type == {"s": 1, "n": 2}[kind[1]]
    assert np.array_equal(t.data, want)


REF_BIN = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "yart_ref")
TEX_KINDS = {"4s": (4, 1, "0,1,2,3"), "2n": (2, 2, "1,2"), "3s": (3, 1, "0,1,2"), "3n": (3, 2, "0,1,2")}


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/yart_ref not built here")
@pytest.mark.parametrize("seed", range(32))
def test_random_images_equal_reference_loadtexture(api, tmp_path, seed):
    """Decoder fuzz: random PNG / JPEG files (gltf_assets.random_image) through the reference's loadTexture (stb_image) and
    through the importer's own decoders — the texture bytes are the same (4060 such files run once: all equal)."""
    import subprocess

    def writejpg(png, quality):
        src, out = os.path.join(tmp_path, "src.png"), os.path.join(tmp_path, "src.jpg")
        with open(src, "wb") as f:
            f.write(png)
        subprocess.run([REF_BIN, "writejpg", src, str(quality), out], check=True)
        return open(out, "rb").read()
    data, ext, desc = ga.random_image(seed, writejpg)
    path = os.path.join(tmp_path, "i." + ext)
    with open(path, "wb") as f:
        f.write(data)
    tag = list(TEX_KINDS)[seed // 2 % len(TEX_KINDS)]
    c, typ, ch = TEX_KINDS[tag]
    want_path = os.path.join(tmp_path, "o.tex")
    subprocess.run([REF_BIN, "texture", path, str(c), str(typ), ch, want_path], check=True)
    b = ga.GltfBuilder()
    mat = b.material(**_material_for(tag, b.texture(b.image(data, **({"mime": "image/jpeg"} if ext == "jpg" else {})))))
    b.node(_triangle_mesh(b, mat), root=True)
    glb = os.path.join(tmp_path, "t.glb")
    b.write_glb(glb)
    s = load(api, glb, tmp_path)
    raw = open(want_path, "rb").read()
    w, h, cc = struct.unpack_from("<3I", raw)
    want = np.frombuffer(raw, np.uint8, offset=12).reshape(h, w, cc)
    assert len(s.textures) == 1, desc
    assert s.textures[0].data.shape == want.shape and np.array_equal(s.textures[0].data, want), desc


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/yart_ref not built here")
@pytest.mark.parametrize("seed", range(12))
def test_random_hdr_files_equal_reference_loadtexturehdr(api, tmp_path, seed):
    """Random Radiance files — 1..59 x 1..39 pixels, RLE and flat scanlines, both magics, RGBE from floats over 10 decades or
    random bytes (every exponent byte, 0 and 255 included) — through the reference's loadTextureHDR and the importer: the
    float texels are the same bit for bit (600 such files run once: all equal)."""
    import subprocess
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 60))
    if rng.random() < 0.5:
        f = np.exp(rng.uniform(-12, 12, (h, w, 3))).astype(np.float32)
        if rng.random() < 0.5:
            f = np.repeat(np.repeat(f[::4, ::4], 4, 0), 4, 1)[:h, :w]
        rgbe = ga.rgbe_from_float(f)
    else:
        rgbe = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
    rle = bool(rng.random() < 0.6) and w >= 8
    hdr = os.path.join(tmp_path, "e.hdr")
    with open(hdr, "wb") as f:
        f.write(ga.hdr_encode(rgbe, rle=rle, magic=b"#?RGBE" if rng.random() < 0.3 else b"#?RADIANCE"))
    want_path = os.path.join(tmp_path, "e.hdrtex")
    subprocess.run([REF_BIN, "hdr", hdr, want_path], check=True)
    b = ga.GltfBuilder()
    b.node(_triangle_mesh(b, b.material()), root=True)
    glb = os.path.join(tmp_path, "t.glb")
    b.write_glb(glb)
    s = load(api, glb, tmp_path, env_hdr=hdr, env_radius=10.0)
    raw = open(want_path, "rb").read()
    ww, hh = struct.unpack_from("<2I", raw)
    want = np.frombuffer(raw, np.float32, offset=8).reshape(hh, ww, 3)
    got = s.textures[-1].data
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, rle)


@pytest.mark.parametrize("name", ["env_rle", "env_flat", "env_tiny"])
def test_hdr_environment_equals_reference_loadtexturehdr(api, tmp_path, name):
    b = ga.GltfBuilder()
    b.node(_triangle_mesh(b, b.material()), root=True)
    glb = os.path.join(tmp_path, "t.glb")
    b.write_glb(glb)
    s = load(api, glb, tmp_path, env_hdr=os.path.join(G, name + ".hdr"), env_radius=42.0)
    with open(os.path.join(G, name + ".hdrtex"), "rb") as f:
        raw = f.read()
    w, h = struct.unpack_from("<2I", raw)
    want = np.frombuffer(raw, np.float32, offset=8).reshape(h, w, 3)
    t = s.textures[-1]
    assert t.data.dtype == np.float32 and t.type == 0
    assert np.array_equal(t.data.view(np.uint32), want.view(np.uint32))
    env = s.lights[-1]
    assert (env.type, env.texture, env.radius) == (2, len(s.textures) - 1, 42.0)
    assert np.array_equal(env.fwd, np.eye(4, dtype=np.float32))


def test_node_transforms_equal_reference_transform_algebra(api, tmp_path):
    """T*R*S, its inverse and node.transform * globalTransform == the reference's float4x4 / Transform classes."""
    rows = [l.split() for l in open(os.path.join(G, "xform.txt")) if l.strip()]
    want = np.fromfile(os.path.join(G, "xform.bin"), np.float32).reshape(len(rows), 4, 4, 4)
    b = ga.GltfBuilder()
    glow = b.material(emissiveFactor=[1.0, 1.0, 1.0])
    children = {i: [] for i in range(len(rows))}
    for i, r in enumerate(rows):
        if int(r[10]) >= 0:
            children[int(r[10])].append(i)
    idx = {}

    def emit(i):                              # glTF children must exist before their parent is written
        kids = [emit(c) for c in children[i]]
        v = [float(x) for x in rows[i][:10]]
        mesh = _triangle_mesh(b, glow)
        idx[i] = mesh                         # one mesh per node: identifies the node in the output
        return b.node(mesh, translation=v[0:3], rotation=v[3:7], scale=v[7:10], children=kids, root=(int(rows[i][10]) < 0))
    emit(0)
    glb = os.path.join(tmp_path, "x.glb")
    b.write_glb(glb)
    s = load(api, glb, tmp_path)
    by_mesh = {n.mesh: n for n in s.nodes if n.mesh >= 0}
    light_by_mesh = {l.mesh: l for l in s.lights}
    assert len(by_mesh) == len(rows) == len(light_by_mesh)
    for i in range(len(rows)):
        n, l = by_mesh[idx[i]], light_by_mesh[idx[i]]
        # == on values: the reference's matrices are read back as M * e_j, which loses the sign of zeros
        assert np.array_equal(n.fwd, want[i, 0]), i
        assert np.array_equal(n.inv, want[i, 1]), i
        assert np.array_equal(l.fwd, want[i, 2]), i
        assert np.array_equal(l.inv, want[i, 3]), i


@pytest.fixture(scope="module")
def gallery(api, tmp_path_factory):
    d = tmp_path_factory.mktemp("gallery")
    return load(api, os.path.join(G, "gallery.glb"), d, env_hdr=os.path.join(G, "env_rle.hdr")), d


def test_gallery_materials(gallery):
    s, _ = gallery
    floor, lamp, glass, coat, leaf, plain = s.materials
    f32 = lambda v: float(np.float32(v))
    # gltf.cpp:68-83
    assert floor.tex_base >= 0 and floor.tex_mr >= 0 and floor.tex_normal >= 0
    assert (floor.roughness, floor.metallic, floor.normal_scale) == (f32(0.9), f32(0.2), f32(0.7))
    assert s.textures[floor.tex_base].data.shape == (32, 32, 4) and s.textures[floor.tex_base].type == 1
    assert s.textures[floor.tex_mr].data.shape == (16, 16, 2) and s.textures[floor.tex_mr].type == 2
    assert s.textures[floor.tex_normal].data.shape == (16, 16, 3) and s.textures[floor.tex_normal].type == 2
    # emissiveFactor * emissive_strength (:136), emissive texture RGB sRGB
    assert tuple(lamp.emission) == (f32(14.0), f32(np.float32(0.8) * np.float32(14.0)), f32(np.float32(0.6) * np.float32(14.0)))
    assert s.textures[lamp.tex_emission].data.shape == (8, 8, 3) and s.textures[lamp.tex_emission].type == 1
    # transmission / ior / volume (:91-102, :149-154)
    assert (glass.transmission, glass.ior) == (f32(0.95), f32(1.45))
    assert s.textures[glass.tex_transmission].data.shape == (8, 8, 1)
    assert tuple(glass.volume_color) == (f32(0.7), f32(0.9), f32(0.8))
    assert glass.volume_density == f32(np.float32(1.0) / np.float32(0.6))
    # clearcoat / anisotropy (:107-119); the image given by URI yields no texture (:32-33)
    assert (coat.clearcoat, coat.clearcoat_roughness, coat.anisotropic, coat.aniso_rotation) == (f32(0.8), f32(0.1), f32(0.6), f32(0.4))
    assert coat.tex_base == -1 and coat.tex_clearcoat == -1
    assert tuple(coat.base) == (f32(0.7), f32(0.1), f32(0.08))
    # glTF defaults
    assert tuple(plain.base) == (1.0, 1.0, 1.0) and (plain.metallic, plain.roughness, plain.ior) == (1.0, 1.0, 1.5)
    assert (plain.clearcoat, plain.clearcoat_roughness, plain.transmission, plain.volume_density) == (0.0, f32(0.03), 0.0, 0.0)
    # thin transmission is forced on for every material (:105)
    assert all(m.thin_transmission for m in s.materials)
    # the leaf texture keeps its alpha channel, gamma-2 re-encoded like the colour channels (texture.hpp:78-84)
    a = s.textures[leaf.tex_base].data[..., 3]
    assert set(np.unique(a)) == {0, 255}


def test_gallery_meshes_nodes_and_lights(gallery):
    s, _ = gallery
    room, lamp, sphere, leaf, dflt = s.meshes
    # two TRIANGLES primitives merged, the LINES primitive skipped; second primitive's indices offset by the first's vertices
    assert len(room.positions) == 8 and len(room.faces) == 4
    assert room.faces[:, 3].tolist() == [0, 0, 3, 3]
    assert room.faces[2:, :3].min() >= 4
    assert np.array_equal(room.positions[4], np.array([-4, 0, -4], np.float32))      # read through byteStride / byteOffset
    assert np.all(lamp.tangents == 0)                                                  # no TANGENT attribute
    assert len(sphere.faces) == 24 * 12 * 2
    u16 = np.round(np.clip(0.5, 0, 1) * 65535) / 65535.0                                # normalised u16 texcoords
    assert np.float32(u16) in sphere.uvs[:, 1]
    assert leaf.faces[:, :3].tolist() == [[0, 1, 2], [3, 4, 5]]                        # generated indices
    assert dflt.faces[0, 3] == 0                                                       # material value_or(0)
    # pre-order node list under an identity root
    assert [n.parent for n in s.nodes] == [-1, 0, 1, 1, 3, 0, 5, 0]
    assert [n.mesh for n in s.nodes] == [-1, 0, 1, 2, 3, -1, 1, 4]
    # one AreaLight per emissive triangle; children before the node; the index restarts in every node (gltf.cpp:299-311)
    area = [l for l in s.lights if l.type == 0]
    assert [(l.mesh, l.tri) for l in area] == [(1, 0), (1, 1), (1, 0), (1, 1)]
    assert lamp.face_light.tolist() == [0, 1]
    assert np.all(room.face_light == -1)
    assert not np.array_equal(area[0].fwd, area[2].fwd)                                # the two instances carry their own transforms
    assert s.lights[-1].type == 2                                                      # environment appended last (main.cpp:83)
    # light transform = node.transform * parentGlobal: composing by hand from the node list gives the same matrices
    from yart_amd.yscn import _matmul32
    fwd = _matmul32(s.nodes[2].fwd, s.nodes[1].fwd)
    assert np.array_equal(area[0].fwd, fwd)
    # the matrix node: scale (0.5, 1, 0.5) then translation (-2, -3, 1)
    want = np.diag([0.5, 1, 0.5, 1]).astype(np.float32); want[:3, 3] = (-2, -3, 1)
    assert np.array_equal(s.nodes[6].fwd, want)


def test_gltf_json_variants_give_the_same_scene(api, tmp_path):
    """.glb, .gltf + external .bin (percent-encoded name) and .gltf + base64 data URI are the same asset."""
    outs = []
    for kind in ("glb", "bin", "uri"):
        b = ga.GltfBuilder()
        rgb = (np.arange(5 * 7 * 3).reshape(5, 7, 3) * 3) % 256
        mat = b.material(pbrMetallicRoughness={"baseColorTexture": {"index": b.texture(b.image(ga.png_encode(rgb, 2)))}},
                         emissiveFactor=[0.5, 0.25, 0.125])
        b.node(_triangle_mesh(b, mat), translation=(1, 2, 3), root=True)
        path = os.path.join(tmp_path, f"a_{kind}." + ("glb" if kind == "glb" else "gltf"))
        if kind == "glb": b.write_glb(path)
        elif kind == "bin": b.write_gltf(path, bin_name="a buffer.bin")
        else: b.write_gltf(path)
        out = os.path.join(tmp_path, kind + ".yscn")
        api.gltf_to_yscn(path, out)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2]


def test_yscn_python_roundtrip(gallery):
    from yart_amd import yscn
    s, d = gallery
    raw = open(os.path.join(d, "out.yscn"), "rb").read()
    assert yscn.Scene.frombytes(raw).tobytes() == raw


def _expect_error(api, path, fragment, tmp_path):
    with pytest.raises(api.YartError) as e:
        api.gltf_to_yscn(path, os.path.join(tmp_path, "e.yscn"))
    assert e.value.code == api.YART_E_IO
    assert fragment in str(e.value), str(e.value)


def test_unsupported_or_broken_assets_fail_loudly(api, tmp_path):
    # arithmetic-coded JPEG (SOF9): refused, not approximated
    b = ga.GltfBuilder()
    jpeg = ga.jpeg_encode(np.full((8, 8, 3), 128, np.uint8)).replace(b"\xFF\xC0", b"\xFF\xC9", 1)
    b.node(_triangle_mesh(b, b.material(pbrMetallicRoughness={"baseColorTexture": {"index": b.texture(b.image(jpeg, mime="image/jpeg"))}})), root=True)
    p = os.path.join(tmp_path, "jpeg.glb"); b.write_glb(p)
    _expect_error(api, p, "arithmetic", tmp_path)
    # truncated JPEG
    b = ga.GltfBuilder()
    jpeg = ga.jpeg_encode(np.full((16, 16, 3), 99, np.uint8))[:200]
    b.node(_triangle_mesh(b, b.material(pbrMetallicRoughness={"baseColorTexture": {"index": b.texture(b.image(jpeg, mime="image/jpeg"))}})), root=True)
    p = os.path.join(tmp_path, "jpegcut.glb"); b.write_glb(p)
    _expect_error(api, p, "jpeg", tmp_path)
    # missing NORMAL: the reference dereferences the missing attribute
    b = ga.GltfBuilder()
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    b.node(b.mesh([{"attributes": {"POSITION": b.accessor(pos, "VEC3"), "TEXCOORD_0": b.accessor(pos[:, :2].copy(), "VEC2")},
                    "material": b.material()}]), root=True)
    p = os.path.join(tmp_path, "nonormal.glb"); b.write_glb(p)
    _expect_error(api, p, "NORMAL", tmp_path)
    # sparse accessor
    b = ga.GltfBuilder()
    m = _triangle_mesh(b, b.material())
    b.doc["accessors"][0]["sparse"] = {"count": 1, "indices": {"bufferView": 0, "componentType": 5121}, "values": {"bufferView": 0}}
    b.node(m, root=True)
    p = os.path.join(tmp_path, "sparse.glb"); b.write_glb(p)
    _expect_error(api, p, "sparse", tmp_path)
    # an extension the reference's parser is not configured for
    b = ga.GltfBuilder()
    b.node(_triangle_mesh(b, b.material()), root=True)
    b.doc["extensionsRequired"] = ["KHR_draco_mesh_compression"]
    p = os.path.join(tmp_path, "draco.glb"); b.write_glb(p)
    _expect_error(api, p, "KHR_draco_mesh_compression", tmp_path)
    # accessor running past its buffer view
    b = ga.GltfBuilder()
    m = _triangle_mesh(b, b.material())
    b.doc["accessors"][0]["count"] = 1000
    b.node(m, root=True)
    p = os.path.join(tmp_path, "overrun.glb"); b.write_glb(p)
    _expect_error(api, p, "exceeds", tmp_path)
    # not a glTF file at all / truncated container
    p = os.path.join(tmp_path, "junk.glb")
    open(p, "wb").write(b"glTF" + struct.pack("<II", 2, 4096) + b"\x10\x00\x00\x00JSON{")
    _expect_error(api, p, "", tmp_path)
    _expect_error(api, os.path.join(tmp_path, "absent.glb"), "cannot open", tmp_path)


@pytest.mark.gpu
def test_gallery_render_equals_reference(api):
    """DeviceScene straight from the .glb (+ .hdr environment) == the reference's render of the imported scene."""
    assert api.lib().yart_hip_device_count() > 0
    p = load_params(os.path.join(G, "gallery.txt"))
    ref = np.fromfile(os.path.join(G, "gallery.f32"), np.float32).reshape(p["size"][1], p["size"][0], 4)
    scene = api.DeviceScene(os.path.join(G, "gallery.glb"), device=0, env_hdr=os.path.join(G, "env_rle.hdr"))
    for flags in (0, 1):
        img, st = scene.render(p, flags=flags)
        e = float(np.sqrt(np.mean((img[..., :3].astype(np.float64) - ref[..., :3]) ** 2)))
        same = float(np.mean(np.all(img.view(np.uint32) == ref.view(np.uint32), axis=-1)))
        print(f"gallery flags={flags}: rmse={e:.3e} identical_pixels={same:.4f}")
        assert e < 1e-3
        assert same > 0.99
    assert float(ref[..., :3].max()) > 0.5 and float(ref[..., :3].std()) > 0.01       # a lit, non-trivial image
    scene.close()
