// yart_oracle.cpp — TEST INFRASTRUCTURE (see yart_oracle.hpp): scene construction and
// the intersection / MIS path-tracing routines of the CPU restatement.
#include "yart_oracle.hpp"

namespace orc {

// Builds the pointer graph the way the reference's loader + main() do
// (gltf/gltf.cpp:319-358, main.cpp:79-84), from the .yscn container.
Scene buildScene(const yscn::SceneFile& sf, const float* lutTables) {
  Scene s;
  s.luts.E = lutTables; s.luts.Eavg = lutTables + 1024; s.luts.baseE = lutTables + 1056;
  s.luts.baseEavg = lutTables + 5152; s.luts.glassE = lutTables + 5408; s.luts.glassInvE = lutTables + 9760;
  s.textures.resize(sf.textures.size());
  for (size_t i = 0; i < sf.textures.size(); i++) {
    const auto& t = sf.textures[i];
    Texture& o = s.textures[i];
    o.w = t.width; o.h = t.height; o.c = t.channels; o.type = t.type; o.isFloat = t.dtype == 1;
    o.u8 = t.u8.data(); o.f32 = t.f32.data();
  }
  auto T = [&](int32_t i) -> const Texture* { return i < 0 ? nullptr : &s.textures[size_t(i)]; };
  for (const auto& m : sf.materials) {                       // parametric.cpp:11-68
    Material o;
    o.base = {m.base[0], m.base[1], m.base[2]};
    o.emission = {m.emission[0], m.emission[1], m.emission[2]};
    o.volumeColor = {m.volumeColor[0], m.volumeColor[1], m.volumeColor[2]};
    o.cTrans = m.transmission; o.cMetallic = m.metallic; o.ior = m.ior; o.roughness = m.roughness;
    o.anisotropic = m.anisotropic; o.clearcoat = m.clearcoat; o.clearcoatRoughness = m.clearcoatRoughness;
    o.volumeDensity = m.volumeDensity; o.thin = m.thinTransmission != 0;
    o.tBase = T(m.texBase); o.tMR = T(m.texMR); o.tTrans = T(m.texTransmission); o.tNormal = T(m.texNormal);
    o.tCoat = T(m.texClearcoat); o.tEmission = T(m.texEmission);
    Material::rotationZ(-m.anisoRotation, o.rot);
    Material::rotationZ(m.anisoRotation, o.invRot);
    if (o.tBase)
      for (size_t k = 3; k < size_t(o.tBase->w) * o.tBase->h * 4; k += 4)
        if (o.tBase->u8[k] < 255) o.hasAlpha = true;
    o.hasEmission = length2(o.emission) > 0.0f;
    o.lut = nullptr;
    s.materials.push_back(o);
  }
  for (auto& m : s.materials) m.lut = &s.luts;
  for (const auto& m : sf.meshes) {                          // mesh.hpp:27-61
    auto o = std::make_unique<Mesh>();
    o->pos.resize(m.nVertices); o->nrm.resize(m.nVertices); o->tan.resize(m.nVertices); o->uv.resize(m.nVertices);
    for (uint32_t v = 0; v < m.nVertices; v++) {
      o->pos[v] = {m.positions[3 * v], m.positions[3 * v + 1], m.positions[3 * v + 2]};
      o->nrm[v] = {m.normals[3 * v], m.normals[3 * v + 1], m.normals[3 * v + 2]};
      o->tan[v] = {m.tangents[4 * v], m.tangents[4 * v + 1], m.tangents[4 * v + 2], m.tangents[4 * v + 3]};
      o->uv[v] = {m.uvs[2 * v], m.uvs[2 * v + 1]};
    }
    o->tri.resize(size_t(m.nFaces) * 3); o->mat.resize(m.nFaces); o->light.resize(m.nFaces);
    for (uint32_t f = 0; f < m.nFaces; f++) {
      for (int k = 0; k < 3; k++) o->tri[3 * f + k] = m.faces[4 * f + k];
      o->mat[f] = m.faces[4 * f + 3];
      o->light[f] = m.faceLight[f];
    }
    o->build();
    s.meshes.push_back(std::move(o));
  }
  // node tree (scene.hpp:11-64): children appended after their own subtrees are complete
  std::vector<std::unique_ptr<Node>> flat(sf.nodes.size());
  for (size_t i = 0; i < sf.nodes.size(); i++) {
    flat[i] = std::make_unique<Node>();
    std::memcpy(flat[i]->xf.m, sf.nodes[i].fwd, 64);
    std::memcpy(flat[i]->xf.inv, sf.nodes[i].inv, 64);
    if (sf.nodes[i].mesh >= 0) {
      flat[i]->mesh = s.meshes[size_t(sf.nodes[i].mesh)].get();
      for (const V3& v : flat[i]->mesh->pos) flat[i]->bounds.expand(v);
    }
  }
  // pre-order file layout: a node's subtree is the following run of deeper nodes, so
  // attaching from the back visits children before parents; children are inserted at the
  // front to keep their original order
  for (size_t i = sf.nodes.size(); i-- > 1;) {
    Node* parent = flat[size_t(sf.nodes[i].parent)].get();
    parent->bounds = Bounds::join(parent->bounds, flat[i]->transformedBounds());
    parent->children.insert(parent->children.begin(), std::move(flat[i]));
  }
  s.root = std::move(flat[0]);
  for (const auto& l : sf.lights) {                          // light.cpp, main.cpp:81-84
    Transform xf;
    std::memcpy(xf.m, l.fwd, 64); std::memcpy(xf.inv, l.inv, 64);
    V3 e{l.emission[0], l.emission[1], l.emission[2]};
    if (l.type == 0) {
      auto a = std::make_unique<AreaLight>(s.meshes[size_t(l.mesh)].get(), l.tri, e, xf);
      a->twoSided = l.twoSided != 0;
      s.lights.push_back(std::move(a));
    } else if (l.type == 1) {
      s.lights.push_back(std::make_unique<UniformInfiniteLight>(l.radius, e));
    } else {
      s.lights.push_back(std::make_unique<ImageInfiniteLight>(l.radius, &s.textures[size_t(l.texture)], xf));
    }
  }
  for (const auto& l : s.lights) {                           // light-sampler.cpp:32-50
    if (l->kind() == Light::Infinite) s.infLights.push_back(l.get());
    else {
      s.areaLights.push_back(l.get());
      s.powers.push_back(s.totalPower + l->power());
      s.totalPower += l->power();
    }
  }
  return s;
}

bool Integrator::testBox(const Ray& ray, float t0, float t1, const Bounds& b, float* d) {   // ray-integrator.cpp:231-261
  auto pick = [&](int which, int axis) { return which == 0 ? b.mn[axis] : b.mx[axis]; };
  V3 bmin(pick(ray.sign[0], 0), pick(ray.sign[1], 1), pick(ray.sign[2], 2));
  V3 bmax(pick(1 - ray.sign[0], 0), pick(1 - ray.sign[1], 1), pick(1 - ray.sign[2], 2));
  V3 tmin = bmin * ray.idir + ray.odir;                       // vec.hpp:325-334: a*b + c, unfused
  V3 tmax = bmax * ray.idir + ray.odir;
  t0 = mmax(tmin.x, t0); t0 = mmax(tmin.y, t0); t0 = mmax(tmin.z, t0);
  t1 = mmin(tmax.x, t1); t1 = mmin(tmax.y, t1); t1 = mmin(tmax.z, t1);
  *d = t0;
  return t1 >= t0;
}

bool Integrator::testTriangle(const Ray& ray, float tMin, Hit& hit, const Mesh& mesh, uint32_t idx) const {
  nTri++; nTriNee += ray.nee;
  const uint32_t i0 = mesh.tri[3 * idx], i1 = mesh.tri[3 * idx + 1], i2 = mesh.tri[3 * idx + 2];
  const V3 p0 = mesh.pos[i0], p1 = mesh.pos[i1], p2 = mesh.pos[i2];
  const V3 e1 = p1 - p0, e2 = p2 - p0;
  const V3 re2 = cross(ray.d, e2);
  const float det = dot(e1, re2);
  bool back = det < 0;
  if (std::abs(det) < 1e-12) return false;                    // double epsilon, math_base.hpp:11
  const float inv = 1.0f / det;
  const V3 b = ray.o - p0;
  const float u = dot(b, re2) * inv;
  if (u < 0.0f || u > 1.0f) return false;
  const V3 be1 = cross(b, e1);
  const float v = dot(ray.d, be1) * inv;
  if (v < 0.0f || u + v > 1.0f) return false;
  const float t = dot(e2, be1) * inv;
  if (t <= tMin || hit.t <= t) return false;
  const Material& bsdf = scene->materials[mesh.mat[idx]];
  const float w = 1.0f - u - v;
  V2 uv = w * mesh.uv[i0] + u * mesh.uv[i1] + v * mesh.uv[i2];
  float alpha = bsdf.alpha(uv);
  if (alpha < 1.0f && sampler->get1D() > alpha) return false;  // consumes a sampler dimension (:207-211)
  hit.n = w * mesh.nrm[i0] + u * mesh.nrm[i1] + v * mesh.nrm[i2];
  if (ray.nee && bsdf.transparent()) {
    hit.attenuation = hit.attenuation * (absDot(hit.n, ray.d) * bsdf.baseAt(uv));
    return false;
  }
  hit.t = t; hit.tg = V3(w, u, v); hit.bsdf = &bsdf; hit.uv = uv;
  hit.p = ray.o + (t * ray.d);
  hit.idx = idx; hit.backSide = back;
  return true;
}

bool Integrator::testBVH(const Ray& ray, float tMin, Hit& hit, const Mesh& mesh) const {
  const BVHNode* node = &mesh.nodes[0];
  const BVHNode* stack[64];
  float dStack[64];
  float d;
  uint32_t sp = 0;
  bool didHit = false;
  nBox++; nBoxNee += ray.nee;
  if (!testBox(ray, tMin, hit.t, node->b, &d)) return false;
  while (true) {
    if (d < hit.t) {
      if (node->span > 0) {
        for (size_t i = 0; i < node->span; i++) {
          uint32_t idx = uint32_t(mesh.idx[node->leftFirst + i]);
          didHit |= testTriangle(ray, tMin, hit, mesh, idx);
          if (ray.nee && didHit) break;
        }
        if (sp == 0) break;
        node = stack[--sp]; d = dStack[sp];
      } else {
        const BVHNode* c1 = &mesh.nodes[node->leftFirst];
        const BVHNode* c2 = &mesh.nodes[node->leftFirst + 1];
        float d1, d2;
        nBox += 2; nBoxNee += 2 * ray.nee;
        bool h1 = testBox(ray, tMin, hit.t, c1->b, &d1);
        bool h2 = testBox(ray, tMin, hit.t, c2->b, &d2);
        if (h1) {
          if (h2) {
            if (d1 > d2) { std::swap(d1, d2); std::swap(c1, c2); }
            dStack[sp] = d2; stack[sp++] = c2;
          }
          node = c1; d = d1;
        } else if (h2) {
          node = c2; d = d2;
        } else {
          if (sp == 0) break;
          node = stack[--sp]; d = dStack[sp];
        }
      }
    } else {
      if (sp == 0) break;
      node = stack[--sp]; d = dStack[sp];
    }
  }
  return didHit;
}

bool Integrator::testMesh(const Ray& ray, float tMin, Hit& hit, const Mesh& mesh) const {
  bool didHit = testBVH(ray, tMin, hit, mesh);
  if (didHit) {
    const uint32_t i0 = mesh.tri[3 * hit.idx], i1 = mesh.tri[3 * hit.idx + 1], i2 = mesh.tri[3 * hit.idx + 2];
    const V4 &t0 = mesh.tan[i0], &t1 = mesh.tan[i1], &t2 = mesh.tan[i2];
    V4 tg;
    tg.x = hit.tg.x * t0.x + hit.tg.y * t1.x + hit.tg.z * t2.x;
    tg.y = hit.tg.x * t0.y + hit.tg.y * t1.y + hit.tg.z * t2.y;
    tg.z = hit.tg.x * t0.z + hit.tg.y * t1.z + hit.tg.z * t2.z;
    tg.w = hit.tg.x * t0.w + hit.tg.y * t1.w + hit.tg.z * t2.w;
    hit.n = hit.bsdf->normal(hit.n, tg, hit.uv);
    if (absDot(hit.n, V3(0, 1, 0)) > 0.999f) hit.tg = V3(1, 0, 0);
    else hit.tg = normalized(cross(hit.n, V3(0, 1, 0)));
    hit.lightIdx = mesh.light[hit.idx];
  }
  return didHit;
}

bool Integrator::testNode(const Ray& ray, float tMin, Hit& hit, const Node& node) const {
  Ray r(node.xf.applyInverse(ray.o, Transform::Point), node.xf.applyInverse(ray.d, Transform::Vector));
  r.nee = ray.nee;
  float d;
  nBox++; nBoxNee += ray.nee;
  if (!testBox(r, tMin, hit.t, node.bounds, &d) || hit.t < d) return false;
  bool didHit = false;
  if (node.mesh) didHit = testMesh(r, tMin, hit, *node.mesh);
  for (const auto& c : node.children) didHit |= testNode(r, tMin, hit, *c);
  if (!didHit) return false;
  hit.p = node.xf.apply(hit.p, Transform::Point);
  hit.n = node.xf.apply(hit.n, Transform::Normal);
  hit.tg = node.xf.apply(hit.tg, Transform::Vector);
  return true;
}

bool Integrator::unoccluded(V3 from, V3 to, V3* att) const {
  Ray r(from, normalized(to - from));
  r.nee = true;
  Hit h;
  h.t = length(to - from) - 0.001f;
  nTrav++; nTravNee++;
  bool occluded = testNode(r, 0.001f, h, *scene->root);
  *att = h.attenuation;
  return !occluded;
}

V3 Integrator::Ld(V3 wo, const Hit& hit) {
  if (scene->lights.empty()) return {};
  float uc = sampler->get1D();
  V2 u = sampler->get2D();
  float pl;
  const Light* l = scene->sampleLight(uc, pl);
  LightSample ls = l->sample(hit.p, u);
  V3 f = hit.bsdf->f(wo, ls.wi, hit.n, hit.tg, hit.uv);
  V3 att(1.0f);
  if (length2(f) == 0.0f || !unoccluded(hit.p, ls.p, &att)) return {};
  rays++;
  float pdfB = hit.bsdf->pdf(wo, ls.wi, hit.n, hit.tg, hit.uv);
  float pdfL = pl * ls.pdf / absDot(ls.n, ls.wi);
  if (l->kind() == Light::Area) pdfL *= length2(hit.p - ls.p);
  return ls.Li * f * att * absDot(ls.wi, hit.n) / (pdfB + pdfL);
}

V3 Integrator::Li(Ray ray) {
  Hit last;
  V3 L(0.0f), att(1.0f);
  uint32_t depth = 0;
  bool spec = false, reg = false;
  float lastPdf = 0.0f, acc = 0.0f;
  while (depth < maxDepth) {
    rays++;
    Hit hit;
    nTrav++;
    bool didHit = testNode(ray, 0.001f, hit, *scene->root);
    if (!didHit) {
      for (const Light* l : scene->infLights) {
        V3 Le = l->Le(octahedralUV(ray.d));
        if (depth == 0 || spec) L = L + att * Le;
        else {
          float pl = l->pdf(ray.d);
          float w = lastPdf / (lastPdf + pl);
          L = L + att * w * Le;
        }
      }
      L = L + att * background;
      break;
    }
    nShade++;
    V2 u = sampler->get2D();
    float uc = sampler->get1D();
    float uc2 = sampler->get1D();
    BSDFSample res = hit.bsdf->sample(-ray.d, hit.n, hit.tg, hit.uv, u, uc, uc2, reg);
    if (res.scatter & Emitted) {
      if (depth == 0 || spec) L = L + att * res.Le;
      else if (hit.lightIdx != -1) {
        const Light& l = *scene->lights[size_t(hit.lightIdx)];
        float pl = l.pdf(-ray.d) * length2(last.p - hit.p) * scene->lightP(size_t(hit.lightIdx)) / absDot(-ray.d, hit.n);
        float w = lastPdf / (lastPdf + pl);
        L = L + att * w * res.Le;
      }
    }
    if (!(res.scatter & (Reflected | Transmitted))) break;
    if (!(res.scatter & (Emitted | Specular))) L = L + att * Ld(-ray.d, hit);
    V3 fcos = res.f * absDot(res.wi, hit.n);
    att = att * (fcos / res.pdf);
    if (hit.backSide) att = att * hit.bsdf->attenuation(hit.t);
    ray = Ray(hit.p, res.wi);
    spec = (res.scatter & Specular) != 0;
    acc += res.roughness;
    reg = acc > 0.5f;
    lastPdf = res.pdf;
    last = hit;
    depth++;
    if (depth > 1 && maxComponent(att) < 1.0f) {
      float q = std::max(0.0f, 1.0f - maxComponent(att));
      if (sampler->get1D() < q) break;
      att = att / (1.0f - q);
    }
  }
  return L;
}

}  // namespace orc
