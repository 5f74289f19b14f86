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

// bucket means sorted by luma: std::sort(first, last, luma(a) < luma(b)) for n <= 16 == insertion sort
YART_HD void bucketMeansSorted(f3* acc, const uint32_t* cnt, int m) {
  for (int i = 0; i < m; i++) acc[i] = acc[i] / float(cnt[i]);
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
}
// the Gini function of the sorted bucket means (estimator.hpp:126-133 / :177-184); leaves their sum in `sum`
YART_HD float giniOfSorted(const f3* acc, int m, f3& sum) {
  f3 weighted = mk3(0);
  sum = mk3(0);
  for (int i = 0; i < m; i++) {
    sum += acc[i];
    weighted += float(i + 1) * acc[i];
  }
  return (2.0f * luma(weighted)) / (float(m) * luma(sum)) - float(m + 1) / float(m);
}

// acc[i] / cnt[i] hold the bucket sums and counts; returns getValue() (estimator.hpp:162-192)
YART_HD f3 gmonFinish(f3* acc, const uint32_t* cnt, int m) {
  if (m == 1) return acc[0] / float(cnt[0]);
  bucketMeansSorted(acc, cnt, m);
  f3 sum;
  float G = giniOfSorted(acc, m, sum);
  if (G > 1.0f) G = 1.0f;
  // size_t(G * float(m/2)): NaN / negative G behave as c = 0 on the reference
  // platform (the 2^63 index wraps to the whole range; see DESIGN.md "GMoN corner")
  float cf = G * float(m / 2);
  int c = (cf > 0.0f) ? int(cf) : 0;
  sum = mk3(0.0f);
  for (int i = c; i < m - c; i++) sum += acc[i];
  return sum / float(m - 2 * c);
}

// The reference's other estimators (core/estimator.hpp:29-141; integrator.cpp:17-18 picks one at compile time,
// GMoN in the shipped source, MeanEstimator in the commented line). Selected by YartRenderParams.estimator.
enum EstimatorKind : int { EST_GMON = 0, EST_MEAN = 1, EST_MON = 2, EST_GMONB = 3 };

YART_HD int estimatorBuckets(int kind, int32_t n) { return kind == EST_MEAN ? 1 : gmonBuckets(n); }   // :58, :99, :151
YART_HD bool estimatorAccepts(int kind, f3 s) {
  if (kind == EST_GMON) return gmonAccepts(s);
  return !((s.x != s.x) || (s.y != s.y) || (s.z != s.z));       // Mean / MoN / GMoNb drop NaN samples only (:35, :62, :103)
}
// nSamples: the wave's sample count (MeanEstimator divides by it, not by the number of accepted samples, :39-41)
YART_HD f3 estimatorFinish(int kind, f3* acc, const uint32_t* cnt, int m, uint32_t nSamples) {
  if (kind == EST_GMON) return gmonFinish(acc, cnt, m);
  if (kind == EST_MEAN) return acc[0] / float(nSamples);
  if (m == 1) return acc[0] / float(cnt[0]);                      // :69, :110
  bucketMeansSorted(acc, cnt, m);
  if (kind == EST_GMONB) {                                        // :124-140
    f3 sum;
    const float G = giniOfSorted(acc, m, sum);
    if (G <= 0.25f) return sum / float(m);
  }
  return acc[m / 2];                                              // median of means (:84, :139)
}

}  // namespace yart_hip
