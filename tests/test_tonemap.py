"""Tonemap / output step: reference core/tonemapping.hpp (AgX, three looks)
as cpu/tile-renderer.hpp:234-240 applies it, and output/ppm.cpp's 8-bit encoding.

Goldens (tests/golden/material.agx_*.{f32,ppm}) were produced by the compiled reference
(`yart_ref tonemap`, tests/golden/make_goldens.py) from its own `material` frame."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import GOLDEN, ROOT

ORACLE = os.path.join(ROOT, "oracle", "_build", "yart_oracle")
W, H = 96, 64
LOOKS = [("none", "none"), ("golden", "golden"), ("punchy", "punchy"), ("-", "raw")]


def _ppm_bytes(path):
    raw = open(path, "rb").read()
    header = b"P6\n%d %d\n255\n" % (W, H)
    assert raw.startswith(header)
    return np.frombuffer(raw[len(header):], np.uint8).reshape(H, W, 3)


@pytest.mark.parametrize("look,tag", LOOKS)
def test_oracle_tonemap_matches_reference_golden(built, tmp_path, look, tag):
    """The CPU restatement (oracle_main.cpp `tonemap`) is pinned bit for bit on the reference's output."""
    f32, ppm = tmp_path / "o.f32", tmp_path / "o.ppm"
    subprocess.run([ORACLE, "tonemap", os.path.join(GOLDEN, "material.f32"), str(W), str(H), look, str(f32), str(ppm)],
                   check=True)
    assert open(ppm, "rb").read() == open(os.path.join(GOLDEN, f"material.agx_{tag}.ppm"), "rb").read()
    if look != "-":
        assert open(f32, "rb").read() == open(os.path.join(GOLDEN, f"material.agx_{tag}.f32"), "rb").read()


@pytest.mark.parametrize("look,tag", LOOKS)
def test_device_headers_tonemap_bit_exact_on_host(hostsim, tmp_path, look, tag):
    """csrc/tonemap.hpp + csrc/libm_pow.hpp (glibc's log2f / powf algorithms) compiled for the host."""
    f32, ppm = tmp_path / "h.f32", tmp_path / "h.ppm"
    subprocess.run([hostsim, "tonemap", os.path.join(GOLDEN, "material.f32"), str(W), str(H), look, str(f32), str(ppm)],
                   check=True)
    assert open(ppm, "rb").read() == open(os.path.join(GOLDEN, f"material.agx_{tag}.ppm"), "rb").read()
    if look != "-":
        assert open(f32, "rb").read() == open(os.path.join(GOLDEN, f"material.agx_{tag}.f32"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("look,tag", LOOKS)
def test_device_tonemap_vs_reference_golden(built, look, tag):
    """k_tonemap_agx + k_encode_rgb8 through the C ABI: log2 / pow follow glibc's algorithms on the
    device (csrc/libm_pow.hpp, exhaustively checked against libm), so the frame and the bytes are
    expected to be the reference's bit for bit; the asserted bar stays north_star's (max abs error 2e-5
    on [0, 1] values, a byte at most one level off) in case the GPU box's glibc differs."""
    from yart_amd import api
    assert api.lib().yart_hip_device_count() > 0
    hdr = np.fromfile(os.path.join(GOLDEN, "material.f32"), np.float32).reshape(H, W, 4)
    ldr, rgb = api.tonemap(hdr, None if look == "-" else look)
    ref_rgb = _ppm_bytes(os.path.join(GOLDEN, f"material.agx_{tag}.ppm"))
    diff = np.abs(rgb.astype(np.int32) - ref_rgb.astype(np.int32))
    print(f"look {tag}: bytes differing {int((diff > 0).sum())} of {diff.size}, max {int(diff.max())}")
    assert diff.max() <= 1 and (diff > 0).mean() <= 1e-3
    if look != "-":
        ref = np.fromfile(os.path.join(GOLDEN, f"material.agx_{tag}.f32"), np.float32).reshape(H, W, 4)
        err = np.abs(ldr.astype(np.float64) - ref.astype(np.float64))
        print(f"look {tag}: max abs error {np.nanmax(err):.3e}, identical floats "
              f"{(ldr.view(np.uint32) == ref.view(np.uint32)).mean():.4f}")
        assert np.all(ldr[..., 3] == 1.0)
        assert np.array_equal(np.isnan(ldr), np.isnan(ref))
        assert np.nanmax(err) <= 2e-5
    else:
        assert np.array_equal(ldr, hdr)


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "yart_ref")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/yart_ref not built here")
@pytest.mark.parametrize("seed", range(6))
def test_tonemap_random_frames_vs_reference(hostsim, tmp_path, seed):
    """Random HDR frames — 11 decades of magnitude, zeros, denormals, negative values, infinities and NaNs (a rendered frame holds
    none of the last three: GMoN drops such samples) — through the compiled reference, the device headers on the host and the
    oracle: the 8-bit output is the same byte for byte and every float is the same bit for bit, except that a NaN may be another
    NaN (negative inputs of the look "none": the NaN the reference's log2 makes of them keeps its sign through min / max there)."""
    rng = np.random.RandomState(seed)
    x = np.exp(rng.uniform(-14, 12, (H, W, 4))).astype(np.float32)
    m = rng.rand(H, W, 4)
    x[m < 0.03] = 0.0
    x[(m >= 0.03) & (m < 0.05)] *= -1
    if seed % 2 == 0:
        x[(m >= 0.05) & (m < 0.055)] = np.inf
        x[(m >= 0.055) & (m < 0.06)] = np.nan
    if seed % 3 == 1:
        x[m > 0.9] = np.float32(1e-42)
    x[..., 3] = 1.0
    src = str(tmp_path / "in.f32")
    x.tofile(src)
    for look in ("none", "golden", "punchy", "-"):
        outs = {}
        for name, exe in (("ref", REF_BIN), ("device headers", hostsim), ("oracle", ORACLE)):
            f32, ppm = str(tmp_path / "o.f32"), str(tmp_path / "o.ppm")
            subprocess.run([exe, "tonemap", src, str(W), str(H), look, f32, ppm], check=True, capture_output=True)
            outs[name] = (open(ppm, "rb").read(), np.fromfile(f32, np.float32) if look != "-" else None)
        for name in ("device headers", "oracle"):
            assert outs[name][0] == outs["ref"][0], (seed, look, name, "8-bit output differs")
            if look != "-":
                a, b = outs[name][1], outs["ref"][1]
                differ = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
                assert not differ.any(), (seed, look, name, int(differ.sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_device_tonemap_random_frames_vs_reference(built, tmp_path, seed):
    """k_tonemap_agx + k_encode_rgb8 on random HDR frames (as above, without the non-finite and negative values a frame cannot
    hold) against the compiled reference run on the GPU box: floats bit for bit, bytes equal."""
    from yart_amd import api
    if not os.path.exists(REF_BIN):
        pytest.fail("oracle/_ref/yart_ref is missing: the GPU tonemap fuzz compares against it")
    rng = np.random.RandomState(100 + seed)
    x = np.exp(rng.uniform(-14, 12, (H, W, 4))).astype(np.float32)
    m = rng.rand(H, W, 4)
    x[m < 0.03] = 0.0
    if seed % 2:
        x[m > 0.9] = np.float32(1e-42)
    x[..., 3] = 1.0
    src = str(tmp_path / "in.f32")
    x.tofile(src)
    for look in ("none", "golden", "punchy"):
        f32, ppm = str(tmp_path / "o.f32"), str(tmp_path / "o.ppm")
        subprocess.run([REF_BIN, "tonemap", src, str(W), str(H), look, f32, ppm], check=True, capture_output=True)
        ref = np.fromfile(f32, np.float32).reshape(H, W, 4)
        ldr, rgb = api.tonemap(x, look)
        same = (ldr.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(ldr) & np.isnan(ref))
        print(f"seed {seed} look {look}: identical floats {same.mean():.6f}, bytes differing {int((rgb != _ppm_bytes(ppm)).sum())}")
        assert same.all(), (seed, look, int((~same).sum()))
        assert np.array_equal(rgb, _ppm_bytes(ppm))
