"""Reader of the oracle/params.hpp "key v0 v1 ..." files into the dict vocabulary used by
yart_amd.scenes / yart_amd.api (test infrastructure)."""


def load_params(path):
    p = {}
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            k, v = t[0], t[1:]
            if k == "size":
                p[k] = (int(v[0]), int(v[1]))
            elif k in ("spp", "first_wave", "max_wave", "tile", "threads", "depth", "aperture_sides"):
                p[k] = int(float(v[0]))
            elif k == "probe_pixels":
                p[k] = [(int(v[i]), int(v[i + 1])) for i in range(0, len(v), 2)]
            elif len(v) == 1:
                p[k] = float(v[0])
            else:
                p[k] = tuple(float(x) for x in v)
    return p
