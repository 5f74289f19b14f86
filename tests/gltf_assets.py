"""Test-side writers of the asset formats the glTF importer reads: PNG, baseline JPEG, Radiance .hdr, GLB / .gltf.

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


# --------------------------------------------------------------------------------------------- JPEG
_ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
           28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
           47, 55, 62, 63]


class _BitWriter:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1)); self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xff
            self.out.append(b)
            if b == 0xff:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)      # pad with 1-bits


def _jpeg_coefficients(pixels, sampling, quant):
    """Quantised DCT coefficients in zigzag order: per component an array (blocks_y, blocks_x, 64) over the
    MCU-padded grid, plus the frame geometry."""
    px = np.asarray(pixels, np.float64)
    H, W, nc = px.shape
    samp = list(sampling)[:nc]
    hmax = max(h for h, v in samp); vmax = max(v for h, v in samp)
    mcu_x = -(-W // (8 * hmax)); mcu_y = -(-H // (8 * vmax))
    k = np.arange(8)
    D = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * np.where(k[:, None] == 0, np.sqrt(1 / 8), np.sqrt(2 / 8))
    coefs = []
    for c, (h, v) in enumerate(samp):
        fx, fy = hmax // h, vmax // v
        cw, ch = -(-W * h // hmax), -(-H * v // vmax)
        full = np.pad(px[..., c], ((0, ch * fy - H), (0, cw * fx - W)), mode="edge")
        sub = full.reshape(ch, fy, cw, fx).mean(axis=(1, 3))          # box-filtered subsampling
        plane = np.pad(sub, ((0, mcu_y * v * 8 - ch), (0, mcu_x * h * 8 - cw)), mode="edge")
        by, bx = plane.shape[0] // 8, plane.shape[1] // 8
        blocks = plane.reshape(by, 8, bx, 8).transpose(0, 2, 1, 3) - 128.0
        q = quant[0 if c == 0 else 1]
        co = np.rint(np.einsum("ij,yxjk,lk->yxil", D, blocks, D) / q).astype(np.int64).reshape(by, bx, 64)
        coefs.append(np.clip(co[..., _ZIGZAG], -1023, 1023))
    geo = dict(H=H, W=W, nc=nc, samp=samp, hmax=hmax, vmax=vmax, mcu_x=mcu_x, mcu_y=mcu_y)
    return coefs, geo


_DC_SYMS = list(range(12))
_AC_SYMS = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)] + [r << 4 for r in range(1, 15)]
_DC_CODE = {s: (i, 4) for i, s in enumerate(_DC_SYMS)}            # 12 codes of 4 bits
_AC_CODE = {s: (i, 8) for i, s in enumerate(_AC_SYMS)}            # 176 codes of 8 bits (EOB0..14, ZRL, run/size)


def _seg(marker, body):
    return bytes((0xFF, marker)) + struct.pack(">H", len(body) + 2) + body


def _jpeg_header(geo, quant, sof, restart, jfif, adobe_transform, component_ids):
    out = bytearray(b"\xFF\xD8")
    if jfif:
        out += _seg(0xE0, b"JFIF\0\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    if adobe_transform is not None:
        out += _seg(0xEE, b"Adobe\0" + bytes((100, 0, 0, 0, 0, adobe_transform)))
    out += _seg(0xFE, b"test asset")
    for t, q in enumerate(quant):
        out += _seg(0xDB, bytes((t,)) + bytes([q] * 64))
    nc, samp = geo["nc"], geo["samp"]
    out += _seg(sof, struct.pack(">BHHB", 8, geo["H"], geo["W"], nc) +
                b"".join(bytes((component_ids[c], (samp[c][0] << 4) | samp[c][1], 0 if c == 0 else 1)) for c in range(nc)))
    out += _seg(0xC4, bytes((0x00,)) + bytes([0, 0, 0, len(_DC_SYMS)] + [0] * 12) + bytes(_DC_SYMS))
    out += _seg(0xC4, bytes((0x10,)) + bytes([0] * 7 + [len(_AC_SYMS)] + [0] * 8) + bytes(_AC_SYMS))
    if restart:
        out += _seg(0xDD, struct.pack(">H", restart))
    return out


def _scan_units(geo, comps):
    """Blocks (component, bx, by) per MCU of a scan: the component's own block grid for a single-component scan,
    interleaved MCUs otherwise."""
    samp, hmax, vmax, W, H = geo["samp"], geo["hmax"], geo["vmax"], geo["W"], geo["H"]
    if len(comps) == 1:
        c = comps[0]
        h, v = samp[c]
        bw = -(-(-(-W * h // hmax)) // 8); bh = -(-(-(-H * v // vmax)) // 8)
        return [[(c, i, j)] for j in range(bh) for i in range(bw)]
    return [[(c, i * samp[c][0] + x, j * samp[c][1] + y) for c in comps for y in range(samp[c][1]) for x in range(samp[c][0])]
            for j in range(geo["mcu_y"]) for i in range(geo["mcu_x"])]


def _put_value(bw, v, s):
    bw.put(v if v >= 0 else v + (1 << s) - 1, s)


def jpeg_encode(pixels, sampling=((1, 1), (1, 1), (1, 1)), quant=(8, 12), restart=0, interleaved=True, jfif=True,
                adobe_transform=None, component_ids=(1, 2, 3), fill_bytes=False):
    """Baseline sequential JPEG (T.81) of an (h, w, 1|3) uint8 array whose channels are stored as given (Y / Cb / Cr,
    or R / G / B for the RGB-id cases): the caller decides what the samples mean, the decoders under test must agree
    on the result. Flat quantisation tables (luma, chroma steps), fixed-length Huffman codes (12 DC categories of
    4 bits, 176 AC symbols of 8 bits): any valid table is as good as another for a decoder test.
    sampling: (h, v) per component; restart: MCUs per restart interval; interleaved=False writes one scan per
    component."""
    coefs, geo = _jpeg_coefficients(pixels, sampling, quant)
    nc = geo["nc"]
    out = _jpeg_header(geo, quant, 0xC0, restart, jfif, adobe_transform, component_ids)

    def encode_block(bw, zz, pred):
        diff = int(zz[0]) - pred
        s = int(abs(diff)).bit_length()
        bw.put(*_DC_CODE[s])
        if s:
            _put_value(bw, diff, s)
        run = 0
        last = max((i for i in range(1, 64) if zz[i]), default=0)
        for i in range(1, last + 1):
            v = int(zz[i])
            if v == 0:
                run += 1
                continue
            while run > 15:
                bw.put(*_AC_CODE[0xF0]); run -= 16
            s = abs(v).bit_length()
            bw.put(*_AC_CODE[(run << 4) | s])
            _put_value(bw, v, s)
            run = 0
        if last < 63:
            bw.put(*_AC_CODE[0x00])
        return int(zz[0])

    def scan(comps):
        nonlocal out
        out += _seg(0xDA, bytes((len(comps),)) + b"".join(bytes((component_ids[c], 0x00)) for c in comps) + bytes((0, 63, 0)))
        bw = _BitWriter()
        pred = {c: 0 for c in comps}
        units = _scan_units(geo, comps)
        rst = 0
        for n, unit in enumerate(units):
            for c, bx, by in unit:
                pred[c] = encode_block(bw, coefs[c][by, bx], pred[c])
            if restart and (n + 1) % restart == 0 and n + 1 < len(units):
                bw.flush()
                out += bw.out + (b"\xFF" if fill_bytes else b"") + bytes((0xFF, 0xD0 + rst % 8))
                bw = _BitWriter(); rst += 1
                pred = {c: 0 for c in comps}
        bw.flush()
        out += bw.out
    if interleaved or nc == 1:
        scan(list(range(nc)))
    else:
        for c in range(nc):
            scan([c])
    out += b"\xFF\xD9"
    return bytes(out)


def jpeg_encode_progressive(pixels, sampling=((1, 1), (1, 1), (1, 1)), quant=(8, 12), restart=0, script=None,
                            jfif=True, component_ids=(1, 2, 3)):
    """Progressive JPEG (T.81 Annex G: spectral selection + successive approximation) with the same tables as
    jpeg_encode. script: list of scans — ("dc", ah, al) over all components interleaved, or ("ac", component, ss, se,
    ah, al); the default refines every coefficient down to bit 0, so the decoded image equals the baseline file's."""
    coefs, geo = _jpeg_coefficients(pixels, sampling, quant)
    nc = geo["nc"]
    if script is None:
        script = [("dc", 0, 1), ("ac", 0, 1, 5, 0, 2)]
        script += [("ac", c, 1, 63, 0, 1) for c in range(1, nc)]
        script += [("ac", 0, 6, 63, 0, 2), ("ac", 0, 1, 63, 2, 1), ("dc", 1, 0)]
        script += [("ac", c, 1, 63, 1, 0) for c in range(nc)]
    out = _jpeg_header(geo, quant, 0xC2, restart, jfif, None, component_ids)

    for sc in script:
        if sc[0] == "dc":
            _, ah, al = sc
            comps, ss, se = list(range(nc)), 0, 0
        else:
            _, c, ss, se, ah, al = sc
            comps = [c]
        out += _seg(0xDA, bytes((len(comps),)) + b"".join(bytes((component_ids[c], 0x00)) for c in comps) +
                    bytes((ss, se, (ah << 4) | al)))
        bw = _BitWriter()
        state = dict(pred={c: 0 for c in comps}, eobrun=0, pending=[])      # pending: buffered correction bits

        def flush_eobrun():
            if state["eobrun"]:
                n = state["eobrun"].bit_length() - 1
                bw.put(*_AC_CODE[n << 4])
                if n:
                    bw.put(state["eobrun"] & ((1 << n) - 1), n)
                state["eobrun"] = 0
            for b in state["pending"]:
                bw.put(b, 1)
            state["pending"] = []

        def block(c, zz):
            if ss == 0:
                if ah == 0:
                    v = int(zz[0]) >> al                                   # point transform: arithmetic shift
                    diff = v - state["pred"][c]; state["pred"][c] = v
                    s = abs(diff).bit_length()
                    bw.put(*_DC_CODE[s])
                    if s:
                        _put_value(bw, diff, s)
                else:
                    bw.put((int(zz[0]) >> al) & 1, 1)
                return
            t = [abs(int(zz[k])) >> al for k in range(64)]                 # point transform: magnitude shift
            if ah == 0:
                run = 0
                for k in range(ss, se + 1):
                    if t[k] == 0:
                        run += 1
                        continue
                    flush_eobrun()
                    while run > 15:
                        bw.put(*_AC_CODE[0xF0]); run -= 16
                    s = t[k].bit_length()
                    bw.put(*_AC_CODE[(run << 4) | s])
                    _put_value(bw, t[k] if zz[k] > 0 else -t[k], s)
                    run = 0
                if run > 0:
                    state["eobrun"] += 1
                    if state["eobrun"] == 0x7FFF:
                        flush_eobrun()
                return
            eob = max((k for k in range(ss, se + 1) if t[k] == 1), default=-1)   # last newly non-zero coefficient
            run = 0
            bits = []
            for k in range(ss, se + 1):
                if t[k] == 0:
                    run += 1
                    continue
                while run > 15 and k <= eob:
                    flush_eobrun()
                    bw.put(*_AC_CODE[0xF0]); run -= 16
                    for b in bits:
                        bw.put(b, 1)
                    bits = []
                if t[k] > 1:
                    bits.append(t[k] & 1)                                  # correction bit of an already non-zero one
                    continue
                flush_eobrun()
                bw.put(*_AC_CODE[(run << 4) | 1])
                bw.put(1 if zz[k] > 0 else 0, 1)
                for b in bits:
                    bw.put(b, 1)
                bits = []
                run = 0
            if run > 0 or bits:
                state["eobrun"] += 1
                state["pending"] += bits
                if state["eobrun"] == 0x7FFF or len(state["pending"]) > 900:
                    flush_eobrun()

        units = _scan_units(geo, comps)
        rst = 0
        for n, unit in enumerate(units):
            for c, bx, by in unit:
                block(c, coefs[c][by, bx])
            if restart and (n + 1) % restart == 0 and n + 1 < len(units):
                flush_eobrun()
                bw.flush()
                out += bw.out + bytes((0xFF, 0xD0 + rst % 8))
                bw = _BitWriter(); rst += 1
                state["pred"] = {c: 0 for c in comps}
        flush_eobrun()
        bw.flush()
        out += bw.out
    out += b"\xFF\xD9"
    return bytes(out)


def random_image(seed, ref_writejpg=None):
    """A random PNG (even seeds) or JPEG (odd seeds) for the decoder fuzz (tests/test_gltf_import.py): PNG colour types 0 / 2 / 3 /
    4 / 6 at every legal depth, sizes 1..40, random filters, Adam7, split IDAT, palettes with tRNS, colour keys; JPEG sizes 1..69,
    1 or 3 components, ten sampling layouts, random quantisation steps, restart intervals, one scan per component, fill bytes,
    JFIF / Adobe markers, progressive files — and, if ref_writejpg(png_bytes, quality) -> jpeg_bytes is given, files written by the
    reference's own encoder (stb_image_write). Returns (bytes, "png" | "jpg", description)."""
    rng = np.random.default_rng(seed)
    if seed % 2 == 0:
        ct = int(rng.choice([0, 2, 3, 4, 6]))
        depth = int(rng.choice({0: [1, 2, 4, 8, 16], 2: [8, 16], 3: [1, 2, 4, 8], 4: [8, 16], 6: [8, 16]}[ct]))
        h, w = int(rng.integers(1, 41)), int(rng.integers(1, 41))
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
        kw = {}
        if ct == 3:
            npal = int(rng.integers(1, (1 << depth) + 1))
            kw["palette"] = rng.integers(0, 256, (npal, 3))
            pix = rng.integers(0, npal, (h, w, 1))
            if rng.random() < 0.5:
                kw["trns"] = bytes(rng.integers(0, 256, int(rng.integers(1, npal + 1))).astype(np.uint8))
        else:
            pix = rng.integers(0, 1 << depth, (h, w, ch))
            if rng.random() < 0.4:
                pix = (np.add.outer(np.arange(h) * 7, np.arange(w) * 3)[..., None] * np.ones(ch, int)) % (1 << depth)
            if ct in (0, 2) and rng.random() < 0.4:
                key = pix[int(rng.integers(h)), int(rng.integers(w))]
                kw["trns"] = b"".join(struct.pack(">H", int(v)) for v in key)
        fl = tuple(int(v) for v in rng.choice([0, 1, 2, 3, 4], int(rng.integers(1, 6))))
        data = png_encode(pix, ct, depth=depth, filters=fl, interlace=bool(rng.random() < 0.4),
                          idat_split=int(rng.choice([0, 0, 7, 50, 1000])), **kw)
        return data, "png", f"png type {ct} depth {depth} {w}x{h} {sorted(kw)} filters {fl}"
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    nc = int(rng.choice([1, 3, 3, 3]))
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 70 * np.sin(xx / rng.uniform(2, 9)) + 40 * np.cos(yy / rng.uniform(2, 9)),
                    128 + 60 * np.sin((xx + yy) / rng.uniform(3, 9)), 128 + 80 * np.cos(xx / 4.0 - yy / 6.0)], -1)
    img = img + rng.normal(0, rng.uniform(0, 40), (h, w, 3))
    if rng.random() < 0.2:
        img = rng.integers(0, 256, (h, w, 3)).astype(float)
    img = np.clip(img, 0, 255).astype(np.uint8)[..., :nc]
    layouts = [((1, 1), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1)), ((1, 2), (1, 1), (1, 1)),
               ((4, 1), (1, 1), (1, 1)), ((1, 4), (1, 2), (1, 1)), ((2, 2), (2, 1), (1, 1)), ((2, 1), (1, 1), (2, 1)),
               ((4, 2), (1, 1), (1, 1)), ((2, 2), (1, 2), (1, 2))]
    samp = layouts[int(rng.integers(len(layouts)))] if nc == 3 else (((1, 1),), ((2, 2),), ((2, 1),))[int(rng.integers(3))]
    q = (int(rng.integers(1, 40)), int(rng.integers(1, 60)))
    rst = int(rng.choice([0, 0, 1, 2, 3, 7]))
    mode = int(rng.integers(3))
    if mode == 0 and nc == 3 and rng.random() < 0.5 and ref_writejpg is not None:
        quality = int(rng.integers(1, 101))
        return ref_writejpg(png_encode(img, 2), quality), "jpg", f"jpg by stb_image_write q{quality} {w}x{h}"
    if mode < 2:
        kw = dict(sampling=samp, quant=q, restart=rst, interleaved=bool(rng.random() < 0.7), fill_bytes=bool(rng.random() < 0.3))
        if nc == 3 and rng.random() < 0.3:
            kw.update(jfif=bool(rng.random() < 0.5), adobe_transform=int(rng.choice([0, 1])))
        return jpeg_encode(img, **kw), "jpg", f"jpg baseline {w}x{h} x{nc} {kw}"
    return (jpeg_encode_progressive(img, sampling=samp, quant=q, restart=rst), "jpg",
            f"jpg progressive {w}x{h} x{nc} {samp} q{q} restart {rst}")
