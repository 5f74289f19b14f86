"""Preview helper: raw RGBA32F framebuffer -> 8-bit PNG (gamma 2.2, clamp)."""
import struct, sys, zlib
import numpy as np

def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\0" + rgb8[y].tobytes() for y in range(h))
    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))

if __name__ == "__main__":
    src, w, h, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    img = np.fromfile(src, np.float32).reshape(h, w, 4)[..., :3]
    img = np.nan_to_num(img, nan=0.0, posinf=1.0)
    img = np.clip(img / (1.0 + img) * 1.6, 0, 1) ** (1 / 2.2)
    write_png(dst, (img * 255).astype(np.uint8))
