// Test harness (not product code): the glTF / GLB importer of the library (csrc/gltf_reader.hpp with its JSON, PNG, JPEG and
// Radiance readers) as a host program, built with AddressSanitizer + UndefinedBehaviorSanitizer by tests/test_sanitizers.py.
//   import_san <asset.glb | asset.gltf> [env.hdr]      exit 0: imported (prints counts), 1: the importer refused the file
// A sanitizer report aborts the process: that is what the tests look for.
#include <cstdio>
#include <exception>
#include <memory>

#include "../../yart_amd/csrc/gltf_reader.hpp"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: import_san asset.glb [env.hdr]\n"); return 2; }
  try {
    auto loaded = yart_hip::gltf::loadGltf(argv[1]);
    if (argc > 2) yart_hip::gltf::addImageEnvironment(*loaded, argv[2], 100.0f);
    const YartSceneDesc& d = loaded->desc;
    std::printf("{\"textures\": %u, \"materials\": %u, \"meshes\": %u, \"nodes\": %u, \"lights\": %u}\n", d.n_textures, d.n_materials,
                d.n_meshes, d.n_nodes, d.n_lights);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "refused: %s\n", e.what());
    return 1;
  }
}
