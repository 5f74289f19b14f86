"""Helpers for comparing known-answer-test (KAT) JSON files.

A KAT file maps section names to integer lists; float sections (``*_f``, ``sampler``,
``ggx*``, ``camera_rays``, ``radiance``, ``gmon``, ``hits_f`` ...) hold IEEE-754
binary32 bit patterns so that equality checks are exact.
"""
from __future__ import annotations

import json

import numpy as np

INT_SECTIONS = {"hash32", "mixbits", "morton", "log2int", "bvh", "hits_i", "bsdf_i", "lights_i",
                "probe_rays"}


def load(path):
    with open(path) as f:
        return json.load(f)


def as_float(bits):
    return np.asarray(bits, dtype=np.uint32).view(np.float32)


def ulp_distance(a_bits, b_bits):
    """Distance in units in the last place between two float bit-pattern arrays
    (NaN vs NaN counts as 0; sign-magnitude mapped to a monotone integer line)."""
    a = np.asarray(a_bits, dtype=np.uint32).astype(np.int64)
    b = np.asarray(b_bits, dtype=np.uint32).astype(np.int64)

    def key(x):
        return np.where(x & 0x80000000, -(x & 0x7FFFFFFF), x)
    fa, fb = as_float(a_bits), as_float(b_bits)
    d = np.abs(key(a) - key(b))
    both_nan = np.isnan(fa) & np.isnan(fb)
    d[both_nan] = 0
    one_nan = np.isnan(fa) ^ np.isnan(fb)
    d[one_nan] = 1 << 40
    return d


def compare(ref, got, sections=None):
    """Returns {section: dict(n, mismatches, max_ulp, first)} for the given sections."""
    out = {}
    for name in (sections or ref.keys()):
        r, g = ref[name], got.get(name)
        if g is None:
            out[name] = dict(n=len(r), mismatches=len(r), max_ulp=None, first=None, missing=True)
            continue
        if len(r) != len(g):
            out[name] = dict(n=len(r), mismatches=max(len(r), len(g)), max_ulp=None, first=None,
                             length=(len(r), len(g)))
            continue
        if len(r) == 0:
            out[name] = dict(n=0, mismatches=0, max_ulp=0, first=None)
            continue
        if name in INT_SECTIONS:
            ra, ga = np.asarray(r, dtype=np.object_), np.asarray(g, dtype=np.object_)
            bad = np.nonzero(ra != ga)[0]
            out[name] = dict(n=len(r), mismatches=int(len(bad)), max_ulp=None,
                             first=int(bad[0]) if len(bad) else None)
        else:
            d = ulp_distance(r, g)
            bad = np.nonzero(d)[0]
            out[name] = dict(n=len(r), mismatches=int(len(bad)), max_ulp=int(d.max()),
                             first=int(bad[0]) if len(bad) else None)
    return out


if __name__ == "__main__":
    import sys
    a, b = load(sys.argv[1]), load(sys.argv[2])
    for k, v in compare(a, b).items():
        extra = ""
        if v["mismatches"] and v.get("first") is not None and k not in INT_SECTIONS:
            i = v["first"]
            extra = f"  first@{i}: ref={as_float([a[k][i]])[0]!r} got={as_float([b[k][i]])[0]!r}"
        elif v["mismatches"] and v.get("first") is not None:
            i = v["first"]
            extra = f"  first@{i}: ref={a[k][i]} got={b[k][i]}"
        print(f"{k:14s} n={v['n']:7d} mismatches={v['mismatches']:7d} max_ulp={v['max_ulp']}{extra}"
              + (f" LENGTH {v['length']}" if "length" in v else "") + (" MISSING" if v.get("missing") else ""))
