"""The C-ABI library loads, exports every symbol include/yart_hip.h declares, validates
its inputs and — without a HIP device — fails loudly instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT, GOLDEN


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "yart_hip.h")).read()
    return sorted(set(re.findall(r"\b(yart_hip_[a-z0-9_]+)\s*\(", hdr)))


def test_exports_match_header(built):
    from yart_amd import api
    lib = ctypes.CDLL(api.LIB_PATH)
    names = declared_symbols()
    assert set(names) == set(api.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.yart_hip_abi_version() == 3


def test_struct_sizes_match_file_records(built):
    from yart_amd import api
    assert ctypes.sizeof(api.MaterialDesc) == 26 * 4
    assert ctypes.sizeof(api.NodeDesc) == 34 * 4
    assert ctypes.sizeof(api.LightDesc) == 41 * 4


def test_no_cpu_fallback_without_device(built):
    """In a GPU-less container scene creation must fail with YART_E_NO_DEVICE."""
    from yart_amd import api
    if api.lib().yart_hip_device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(api.YartError) as e:
        api.DeviceScene(os.path.join(GOLDEN, "cornell.yscn"))
    assert e.value.code == api.YART_E_NO_DEVICE


def test_null_arguments_are_rejected(built):
    from yart_amd import api
    L = api.lib()
    assert L.yart_hip_scene_create(None, 0, None) == api.YART_E_INVALID
    assert L.yart_hip_render(None, None, None, None, None) == api.YART_E_INVALID
    assert b"null" in L.yart_hip_last_error()


def test_product_does_not_reference_oracle():
    """Only tests/, bench.py's cpu_baseline and smoke() may touch oracle/."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "yart_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hpp", ".hip", ".cpp", ".h", "Makefile")):
            text = open(path, errors="ignore").read()
            for line in text.splitlines():
                s = line.strip()
                if s.startswith(("//", "#", "*", '"""')) or "oracle/" not in s:
                    continue
                assert "include" not in s and "import" not in s and "subprocess" not in s, (path, s)


def test_cpp_mirror_header_compiles_and_links(built, tmp_path):
    """include/yart_hip.hpp (the C++ face of the ABI: DeviceScene / HipTileRenderer with the reference's knob
    names) compiles as C++17 and links against the library."""
    import subprocess
    src = os.path.join(tmp_path, "m.cpp")
    with open(src, "w") as f:
        f.write('#include "yart_hip.hpp"\n'
                "int main() { yart::hip::Buffer b(4, 4); YartCameraDesc c{}; yart::hip::HipTileRenderer r(std::move(b), c);\n"
                "  r.estimator = YART_ESTIMATOR_MON; r.tonemapLook = 1; r.samples = 4;\n"
                "  try { auto s = yart::hip::DeviceScene::fromGltf(\"/nonexistent.glb\"); (void)s; return 2; }\n"
                "  catch (const yart::hip::Error&) {}\n"
                "  return yart_hip_abi_version() == YART_HIP_ABI_VERSION ? 0 : 1; }\n")
    exe = os.path.join(tmp_path, "m")
    lib_dir = os.path.join(ROOT, "yart_amd")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + lib_dir, "-lyart_hip",
                    "-Wl,-rpath," + lib_dir, "-lpthread"], check=True)
    assert subprocess.run([exe]).returncode == 0


def test_crafted_scene_files_are_refused(built, tmp_path):
    """.yscn loader (csrc/scene_file.hpp): sizes come from the file, so products of header fields are bounded before they
    are formed and every read is checked against what is left of the file — a header whose width * height * channels wraps
    to a small number must not get through."""
    import struct
    from yart_amd import api
    L = api.lib()
    head = b"YSCN0001" + struct.pack("<8I", 1, 0, 0, 0, 0, 0, 0, 0)
    cases = {"wrapping texture size": head + struct.pack("<5I", 0x80000001, 0x80000001, 4, 0, 0) + b"\0" * 64,
             "five channels": head + struct.pack("<5I", 4, 4, 5, 0, 0) + b"\0" * 128,
             "truncated texel data": head + struct.pack("<5I", 64, 64, 4, 0, 0) + b"\0" * 100,
             "huge mesh": b"YSCN0001" + struct.pack("<8I", 0, 0, 1, 0, 0, 0, 0, 0) + struct.pack("<2I", 0xffffffff, 0xffffffff),
             "not a scene": b"GARBAGE!" + b"\0" * 64}
    for name, blob in cases.items():
        path = os.path.join(tmp_path, "bad.yscn")
        with open(path, "wb") as f:
            f.write(blob)
        h = ctypes.c_void_p()
        assert L.yart_hip_scene_load(path.encode(), 0, ctypes.byref(h)) == api.YART_E_IO, name
