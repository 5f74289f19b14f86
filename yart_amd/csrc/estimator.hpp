// estimator.hpp — GMoN (Gini median-of-means) per-pixel estimator.
// Restates reference core/estimator.hpp:148-198 as used by cpu/integrator.cpp:15-25
// (mMax = 15), including the accumulation order inside each bucket (sample k goes
// to bucket k mod m, in increasing k) and libstdc++'s std::sort behaviour for
// m <= 16 elements (a plain insertion sort: bits/stl_algo.h __insertion_sort).
#pragma once
#include "ymath.hpp"

namespace yart_hip {

constexpr int kGmonMax = 15;

YART_HD int gmonBuckets(int32_t n) {                   // estimator.hpp:150-152
  int32_t v = 1 + 2 * ((n - 5) / 10);
  if (v < 1) v = 1;
  return v < kGmonMax ? v : kGmonMax;
}
YART_HD float luma(f3 v) { return dot(v, mk3(0.2126f, 0.7152f, 0.0722f)); }   // :19-22

YART_HD bool gmonAccepts(f3 s) {                       // estimator.hpp:154-157
  bool nan = (s.x != s.x) || (s.y != s.y) || (s.z != s.z);
  return !nan && s.x >= 0.0f && s.y >= 0.0f && s.z >= 0.0f;
}

// acc[i] / cnt[i] hold the bucket sums and counts; returns getValue() (estimator.hpp:162-192)
YART_HD f3 gmonFinish(f3* acc, const uint32_t* cnt, int m) {
  if (m == 1) return acc[0] / float(cnt[0]);
  for (int i = 0; i < m; i++) acc[i] = acc[i] / float(cnt[i]);
  // std::sort(first, last, luma(a) < luma(b)) for n <= 16 == insertion sort
  for (int i = 1; i < m; i++) {
    f3 val = acc[i];
    float lv = luma(val);
    if (lv < luma(acc[0])) {
      for (int j = i; j > 0; j--) acc[j] = acc[j - 1];
      acc[0] = val;
    } else {
      int j = i;
      while (lv < luma(acc[j - 1])) { acc[j] = acc[j - 1]; j--; }
      acc[j] = val;
    }
  }
  f3 sum = mk3(0), weighted = mk3(0);
  for (int i = 0; i < m; i++) {
    sum += acc[i];
    weighted += float(i + 1) * acc[i];
  }
  float G = (2.0f * luma(weighted)) / (float(m) * luma(sum)) - float(m + 1) / float(m);
  if (G > 1.0f) G = 1.0f;
  // size_t(G * float(m/2)): NaN / negative G behave as c = 0 on the reference
  // platform (the 2^63 index wraps to the whole range; see DESIGN.md "GMoN corner")
  float cf = G * float(m / 2);
  int c = (cf > 0.0f) ? int(cf) : 0;
  sum = mk3(0.0f);
  for (int i = c; i < m - c; i++) sum += acc[i];
  return sum / float(m - 2 * c);
}

}  // namespace yart_hip
