"""Test-side writers of the asset formats the glTF importer reads: PNG, Radiance .hdr, GLB / .gltf.

TEST INFRASTRUCTURE (used by tests/golden/make_gltf_goldens.py and tests/test_gltf_import.py); written from
the format specifications. Nothing in the product imports this.
"""
import base64
import json
import struct
import zlib

import numpy as np


# ---------------------------------------------------------------------------------------------- PNG
def _chunk(tag: bytes, body: bytes) -> bytes:
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _filter_rows(rows, bpp, filters):
    """rows: list of bytes per scanline; filters: iterable of filter types cycled over the rows."""
    out = bytearray()
    prior = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        out.append(ft)
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prior[i]
            c = prior[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            out.append((v - pred) & 0xff)
        prior = row
    return bytes(out)


def _pack_samples(samples, depth):
    """samples: (h, w*channels) integer array -> list of scanline bytes at the given bit depth."""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(int(v) & 0xff for v in r))
        elif depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        else:
            bits = 0; n = 0; b = bytearray()
            for v in r:
                bits = (bits << depth) | int(v); n += depth
                if n == 8:
                    b.append(bits); bits = 0; n = 0
            if n:
                b.append(bits << (8 - n))
            rows.append(bytes(b))
    return rows


def png_encode(pixels, color_type, depth=8, palette=None, trns=None, filters=(0, 1, 2, 3, 4), interlace=False,
               idat_split=0):
    """pixels: (h, w, channels) integer array of samples at `depth` bits (palette indices for colour type 3).
    trns: bytes of the tRNS chunk. idat_split > 0 cuts the compressed stream into IDAT chunks of that size."""
    pixels = np.asarray(pixels)
    h, w, ch = pixels.shape
    assert ch == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    bits = ch * depth
    bpp = max(1, bits // 8)
    raw = bytearray()
    if not interlace:
        passes = [(0, 0, 1, 1)]
    else:
        passes = list(zip((0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)))
    for x0, y0, dx, dy in passes:
        sub = pixels[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        raw += _filter_rows(_pack_samples(sub.reshape(sub.shape[0], -1), depth), bpp, filters)
    comp = zlib.compress(bytes(raw), 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, int(interlace)))
    if palette is not None:
        out += _chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    out += _chunk(b"tEXt", b"Comment\0test asset")
    step = idat_split or len(comp)
    for i in range(0, len(comp), step):
        out += _chunk(b"IDAT", comp[i:i + step])
    return out + _chunk(b"IEND", b"")


# -------------------------------------------------------------------------------------- Radiance .hdr
def rgbe_from_float(rgb):
    """(h, w, 3) float -> (h, w, 4) uint8 RGBE (mantissas truncated, as Radiance's writers do)."""
    rgb = np.asarray(rgb, np.float64)
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    ok = m > 1e-32
    e = np.zeros(m.shape, np.int64)
    fr = np.zeros(m.shape)
    fr[ok], e[ok] = np.frexp(m[ok])
    scale = np.zeros(m.shape)
    scale[ok] = fr[ok] * 256.0 / m[ok]
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(ok, e + 128, 0).astype(np.uint8)
    out[~ok] = 0
    return out


def hdr_encode(rgbe, rle=True, magic=b"#?RADIANCE"):
    """(h, w, 4) uint8 RGBE -> Radiance file; new-style per-channel RLE scanlines when rle (8 <= w < 32768)."""
    h, w, _ = rgbe.shape
    out = bytearray(magic + b"\n# test asset\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + b"-Y %d +X %d\n" % (h, w))
    for y in range(h):
        if not rle:
            out += rgbe[y].tobytes()
            continue
        out += bytes((2, 2, w >> 8, w & 255))
        for k in range(4):
            row = rgbe[y, :, k]
            i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and row[i + run] == row[i]:
                    run += 1
                if run >= 3:
                    out += bytes((128 + run, int(row[i]))); i += run
                else:
                    j = i
                    while j < w and j - i < 128:
                        if j + 2 < w and row[j] == row[j + 1] == row[j + 2]:
                            break
                        j += 1
                    j = max(j, i + 1)
                    out += bytes((j - i,)) + row[i:j].tobytes(); i = j
    return bytes(out)


# --------------------------------------------------------------------------------------------- glTF
class GltfBuilder:
    """Accumulates buffers / accessors / images and writes a .glb or a .gltf (+ .bin or data URI)."""

    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"version": "2.0", "generator": "yart_amd tests"}, "bufferViews": [], "accessors": [],
                    "images": [], "textures": [], "materials": [], "meshes": [], "nodes": [], "scenes": [{"nodes": []}],
                    "scene": 0, "extensionsUsed": []}

    def _view(self, data: bytes, stride=None, align=4):
        while len(self.bin) % align:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        self.bin += data
        self.doc["bufferViews"].append(v)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, array, type_, component=None, normalized=False, stride=None, byte_offset=0, min_max=False):
        a = np.ascontiguousarray(array)
        ct = component or {np.dtype(np.float32): 5126, np.dtype(np.uint32): 5125, np.dtype(np.uint16): 5123,
                           np.dtype(np.uint8): 5121, np.dtype(np.int16): 5122, np.dtype(np.int8): 5120}[a.dtype]
        ncomp = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}[type_]
        count = a.size // ncomp
        elem = a.dtype.itemsize * ncomp
        if stride:                       # interleave with padding to exercise byteStride
            raw = bytearray(byte_offset + stride * count)
            flat = a.reshape(count, ncomp)
            for i in range(count):
                raw[byte_offset + i * stride: byte_offset + i * stride + elem] = flat[i].tobytes()
            view = self._view(bytes(raw), stride=stride)
        else:
            view = self._view(bytes(byte_offset) + a.tobytes())
        acc = {"bufferView": view, "componentType": ct, "count": count, "type": type_}
        if byte_offset:
            acc["byteOffset"] = byte_offset
        if normalized:
            acc["normalized"] = True
        if min_max:
            flat = a.reshape(count, ncomp).astype(np.float64)
            acc["min"] = flat.min(0).tolist(); acc["max"] = flat.max(0).tolist()
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def image(self, data: bytes, mime="image/png", uri=None):
        if uri is not None:
            self.doc["images"].append({"uri": uri})
        else:
            self.doc["images"].append({"bufferView": self._view(data), "mimeType": mime})
        return len(self.doc["images"]) - 1

    def texture(self, image):
        self.doc["textures"].append({"source": image} if image is not None else {})
        return len(self.doc["textures"]) - 1

    def material(self, **m):
        for ext in m.get("extensions", {}):
            if ext not in self.doc["extensionsUsed"]:
                self.doc["extensionsUsed"].append(ext)
        self.doc["materials"].append(m)
        return len(self.doc["materials"]) - 1

    def mesh(self, primitives):
        self.doc["meshes"].append({"primitives": primitives})
        return len(self.doc["meshes"]) - 1

    def node(self, mesh=None, translation=None, rotation=None, scale=None, matrix=None, children=(), root=False):
        n = {}
        if mesh is not None: n["mesh"] = mesh
        if translation is not None: n["translation"] = [float(v) for v in translation]
        if rotation is not None: n["rotation"] = [float(v) for v in rotation]
        if scale is not None: n["scale"] = [float(v) for v in scale]
        if matrix is not None: n["matrix"] = [float(v) for v in matrix]
        if children: n["children"] = list(children)
        self.doc["nodes"].append(n)
        idx = len(self.doc["nodes"]) - 1
        if root:
            self.doc["scenes"][0]["nodes"].append(idx)
        return idx

    def _json(self, buffer_entry):
        doc = {k: v for k, v in self.doc.items() if v not in ([], {})}
        doc["buffers"] = [buffer_entry]
        return json.dumps(doc, separators=(",", ":")).encode()

    def write_glb(self, path):
        while len(self.bin) % 4:
            self.bin.append(0)
        js = self._json({"byteLength": len(self.bin)})
        js += b" " * ((4 - len(js) % 4) % 4)
        total = 12 + 8 + len(js) + 8 + len(self.bin)
        with open(path, "wb") as f:
            f.write(struct.pack("<4sII", b"glTF", 2, total))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(self.bin), 0x004E4942) + bytes(self.bin))

    def write_gltf(self, path, bin_name=None):
        """bin_name: file name of an external .bin next to `path`; None embeds the buffer as a base64 data URI."""
        if bin_name:
            import os
            with open(os.path.join(os.path.dirname(path), bin_name), "wb") as f:
                f.write(bytes(self.bin))
            entry = {"byteLength": len(self.bin), "uri": bin_name.replace(" ", "%20")}
        else:
            entry = {"byteLength": len(self.bin),
                     "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()}
        with open(path, "wb") as f:
            f.write(self._json(entry))


def quat_axis_angle(axis, angle):
    a = np.asarray(axis, np.float64); a = a / np.linalg.norm(a)
    s = np.sin(angle / 2)
    return [float(np.float32(a[0] * s)), float(np.float32(a[1] * s)), float(np.float32(a[2] * s)), float(np.float32(np.cos(angle / 2)))]
