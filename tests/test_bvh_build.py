"""Host BVH build (csrc/bvh_build.hpp, reference core/bvh.hpp:41-184, 273-347): the task-parallel build gives the
node array and index permutation of the plain recursion byte for byte (which the KAT tests compare with the
reference's). hostsim `bvhcheck` builds every mesh of a scene both ways."""
import json
import os
import subprocess

import pytest

from tests.conftest import GOLDEN


@pytest.mark.parametrize("threads", [2, 5, 8])
def test_parallel_build_equals_serial_on_goldens(hostsim, threads):
    for case in ("cornell", "material"):
        r = subprocess.run([hostsim, "bvhcheck", os.path.join(GOLDEN, case + ".yscn"), str(threads)], check=True,
                           capture_output=True, text=True)
        assert json.loads(r.stdout)["bvhcheck"] == "ok"


def test_parallel_build_equals_serial_on_a_large_mesh(hostsim, tmp_path):
    """264 k triangles in two meshes: deep enough for the split / pool / assemble passes to all take part."""
    from yart_amd import scenes
    s, _ = scenes.sponza_class(64, 64, 1, 2, tex=32, sky=32)
    path = os.path.join(tmp_path, "s.yscn")
    s.save(path)
    r = subprocess.run([hostsim, "bvhcheck", path, "8"], check=True, capture_output=True, text=True)
    out = json.loads(r.stdout)
    assert out["bvhcheck"] == "ok" and out["triangles"] > 200000
    print(out)
