// scene_file.hpp — reader of the .yscn scene container for the product library
// (yart_hip_scene_load). The container stands where the reference's glTF loader
// output stands (src/gltf/gltf.cpp:319-358); layout documented in oracle/yscn.hpp
// and written by yart_amd/yscn.py.
#pragma once
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/yart_hip.h"

namespace yart_hip {

struct LoadedScene {
  std::vector<std::vector<uint8_t>> blobs;      // owns every array the descriptors point into
  std::vector<YartTextureDesc> textures;
  std::vector<YartMaterialDesc> materials;
  std::vector<YartMeshDesc> meshes;
  std::vector<YartNodeDesc> nodes;
  std::vector<YartLightDesc> lights;
  YartSceneDesc desc{};
  // point desc at the vectors (after any of them changed)
  void refreshDesc() {
    desc.n_textures = uint32_t(textures.size()); desc.n_materials = uint32_t(materials.size());
    desc.n_meshes = uint32_t(meshes.size()); desc.n_nodes = uint32_t(nodes.size()); desc.n_lights = uint32_t(lights.size());
    desc.textures = textures.data(); desc.materials = materials.data(); desc.meshes = meshes.data();
    desc.nodes = nodes.data(); desc.lights = lights.data();
  }
};

inline std::unique_ptr<LoadedScene> loadSceneFile(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open scene file " + path);
  std::vector<uint8_t> file;
  std::fseek(f, 0, SEEK_END);
  long sz = std::ftell(f);
  if (sz < 0) { std::fclose(f); throw std::runtime_error("cannot size scene file " + path); }
  std::fseek(f, 0, SEEK_SET);
  file.resize(size_t(sz));
  size_t got = std::fread(file.data(), 1, file.size(), f);
  std::fclose(f);
  if (got != file.size()) throw std::runtime_error("short read on " + path);

  size_t pos = 0;
  // every size below comes from the file: compared without forming pos + n (which could wrap)
  auto need = [&](size_t n) { if (pos > file.size() || n > file.size() - pos) throw std::runtime_error("truncated scene file"); };
  auto u32 = [&]() { need(4); uint32_t v; std::memcpy(&v, &file[pos], 4); pos += 4; return v; };
  auto s = std::make_unique<LoadedScene>();
  auto blob = [&](size_t bytes) -> const void* {
    need(bytes);
    s->blobs.emplace_back(file.begin() + long(pos), file.begin() + long(pos + bytes));
    pos += bytes + (4 - bytes % 4) % 4;
    return s->blobs.back().data();
  };
  need(8);
  if (std::memcmp(file.data(), "YSCN0001", 8) != 0) throw std::runtime_error("not a .yscn file");
  pos = 8;
  uint32_t nt = u32(), nm = u32(), nme = u32(), nn = u32(), nl = u32();
  u32(); u32(); u32();
  for (uint32_t i = 0; i < nt; i++) {
    YartTextureDesc t{};
    t.width = u32(); t.height = u32(); t.channels = u32(); t.is_float = u32(); t.type = u32();
    // bounded before they are multiplied: a crafted header must not wrap the product into a small blob that passes need()
    if (t.width == 0 || t.height == 0 || t.width > (1u << 16) || t.height > (1u << 16) || t.channels == 0 || t.channels > 4)
      throw std::runtime_error("scene file: texture dimensions out of range");
    t.data = blob(size_t(t.width) * t.height * t.channels * (t.is_float ? 4 : 1));
    s->textures.push_back(t);
  }
  static_assert(sizeof(YartMaterialDesc) == 26 * 4, "YartMaterialDesc must match the file record");
  for (uint32_t i = 0; i < nm; i++) {
    need(sizeof(YartMaterialDesc));
    YartMaterialDesc m;
    std::memcpy(&m, &file[pos], sizeof(m)); pos += sizeof(m);
    s->materials.push_back(m);
  }
  for (uint32_t i = 0; i < nme; i++) {
    YartMeshDesc m{};
    m.n_vertices = u32(); m.n_faces = u32();
    if (m.n_vertices > (1u << 28) || m.n_faces > (1u << 28)) throw std::runtime_error("scene file: mesh size out of range");
    m.positions = static_cast<const float*>(blob(size_t(m.n_vertices) * 12));
    m.normals = static_cast<const float*>(blob(size_t(m.n_vertices) * 12));
    m.tangents = static_cast<const float*>(blob(size_t(m.n_vertices) * 16));
    m.uvs = static_cast<const float*>(blob(size_t(m.n_vertices) * 8));
    m.faces = static_cast<const uint32_t*>(blob(size_t(m.n_faces) * 16));
    m.face_light = static_cast<const int32_t*>(blob(size_t(m.n_faces) * 4));
    s->meshes.push_back(m);
  }
  static_assert(sizeof(YartNodeDesc) == 34 * 4 && sizeof(YartLightDesc) == 41 * 4, "record layouts");
  for (uint32_t i = 0; i < nn; i++) {
    need(sizeof(YartNodeDesc));
    YartNodeDesc n;
    std::memcpy(&n, &file[pos], sizeof(n)); pos += sizeof(n);
    s->nodes.push_back(n);
  }
  for (uint32_t i = 0; i < nl; i++) {
    need(sizeof(YartLightDesc));
    YartLightDesc l;
    std::memcpy(&l, &file[pos], sizeof(l)); pos += sizeof(l);
    s->lights.push_back(l);
  }
  s->refreshDesc();
  return s;
}

// the same container, written (byte for byte what yart_amd/yscn.py Scene.tobytes() produces for the same scene)
inline void saveSceneFile(const YartSceneDesc& d, const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot create scene file " + path);
  bool ok = true;
  auto put = [&](const void* p, size_t n) { ok = ok && (n == 0 || std::fwrite(p, 1, n, f) == n); };
  auto u32 = [&](uint32_t v) { put(&v, 4); };
  auto arr = [&](const void* p, size_t n) { static const char zero[4] = {0, 0, 0, 0}; put(p, n); put(zero, (4 - n % 4) % 4); };
  put("YSCN0001", 8);
  u32(d.n_textures); u32(d.n_materials); u32(d.n_meshes); u32(d.n_nodes); u32(d.n_lights); u32(0); u32(0); u32(0);
  for (uint32_t i = 0; i < d.n_textures; i++) {
    const YartTextureDesc& t = d.textures[i];
    u32(t.width); u32(t.height); u32(t.channels); u32(t.is_float); u32(t.type);
    arr(t.data, size_t(t.width) * t.height * t.channels * (t.is_float ? 4 : 1));
  }
  for (uint32_t i = 0; i < d.n_materials; i++) put(&d.materials[i], sizeof(YartMaterialDesc));
  for (uint32_t i = 0; i < d.n_meshes; i++) {
    const YartMeshDesc& m = d.meshes[i];
    u32(m.n_vertices); u32(m.n_faces);
    arr(m.positions, size_t(m.n_vertices) * 12); arr(m.normals, size_t(m.n_vertices) * 12);
    arr(m.tangents, size_t(m.n_vertices) * 16); arr(m.uvs, size_t(m.n_vertices) * 8);
    arr(m.faces, size_t(m.n_faces) * 16);
    if (m.face_light) arr(m.face_light, size_t(m.n_faces) * 4);
    else { const std::vector<int32_t> none(m.n_faces, -1); arr(none.data(), size_t(m.n_faces) * 4); }   // the ABI allows NULL: no lights
  }
  for (uint32_t i = 0; i < d.n_nodes; i++) put(&d.nodes[i], sizeof(YartNodeDesc));
  for (uint32_t i = 0; i < d.n_lights; i++) put(&d.lights[i], sizeof(YartLightDesc));
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) throw std::runtime_error("write error on " + path);
}

}  // namespace yart_hip
