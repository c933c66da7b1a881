"""Known answers for the tiny-cuda-nn half of the path that come from a source INDEPENDENT of oracle/tcnn_oracle.py and of the product:
scipy's spherical harmonics, integer arithmetic written out here, and literal level tables.  Shared by the CPU tests (which hold the
oracle to them) and the GPU tests (which hold libngp_hip.so to them).  Test infrastructure."""
import math

import numpy as np

# ---- SH degree 4 ---------------------------------------------------------------------------------------------------------------------
# tiny-cuda-nn's spherical_harmonics.h lists the real basis in the order l^2 + l + m, m = -l..l, WITH the Condon-Shortley phase
# (-0.4886 y, 0.4886 z, -0.4886 x for l = 1).  From the complex Y_l^m (scipy: CS phase included):
#     m < 0: sqrt(2) Im Y_l^|m|      m = 0: Re Y_l^0      m > 0: sqrt(2) Re Y_l^m


def sh4_scipy(d):
    """(N,3) unit vectors (float64) -> (N,16) real SH values from scipy.special (associated Legendre recursion: nothing of the
    polynomial forms in spherical_harmonics.h / oracle/tcnn_oracle.py:sh4 is used)."""
    from scipy import special
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    theta = np.arccos(np.clip(z, -1.0, 1.0))            # polar
    phi = np.arctan2(y, x)                              # azimuth
    out = np.empty((d.shape[0], 16))
    for l in range(4):
        for m in range(-l, l + 1):
            if hasattr(special, "sph_harm_y"):
                Y = special.sph_harm_y(l, abs(m), theta, phi)
            else:                                       # older scipy: sph_harm(m, l, azimuth, polar)
                Y = special.sph_harm(abs(m), l, phi, theta)
            out[:, l * l + l + m] = Y.real if m == 0 else math.sqrt(2.0) * (Y.imag if m < 0 else Y.real)
    return out


def unit_directions(n, seed=0):
    g = np.random.RandomState(seed)
    d = g.randn(n, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:6] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]]      # poles and axes
    return d


# ---- spatial hash ----------------------------------------------------------------------------------------------------------------------
# grid.h: index = (x * 1) ^ (y * 2654435761) ^ (z * 805459861) in uint32 arithmetic, then % hashmap size (2^19 in the reference's
# configuration, models/networks.py:41).  Worked by hand for (0,1,0): 2654435761 = 0x9E3779B1, low 19 bits 0x779B1 = 489905;
# for (0,0,1): 805459861 = 0x30025795, low 19 bits 0x25795 = 153493; (1,1,1): 1 ^ 0x9E3779B1 ^ 0x30025795 = 0xAE352E25 -> 0x52E25 = 339493.
# (cell x, y, z) -> index within a 2^19-entry level
HASH_KAT = [((0, 0, 0), 0), ((1, 0, 0), 1), ((0, 1, 0), 489905), ((0, 0, 1), 153493), ((1, 1, 1), 339493), ((5, 9, 200), 469844),
            ((1023, 1023, 1023), 308699), ((338, 7, 12345), 336552)]

# ---- level tables ----------------------------------------------------------------------------------------------------------------------
# GridEncodingTemplated's constructor for the reference's configuration (L = 16, F = 2, T = 2^19, N_min = 16,
# b = exp(ln(2048 scale / 16) / 15), models/networks.py:33-48): resolution_l = ceil(b^l * 16 - 1) + 1, size_l = min(roundup8(res^3), 2^19).
#   "float32": b^l evaluated as exp2f(l * log2f(b)) like grid.h -- lands a few 1e-6 ABOVE the integers at levels 5, 10, 15 for scale 0.5
#              (log2f(1.3195079f) = 0.40000004), hence 65 / 257 / 1025;
#   "exact":   exact arithmetic (b^5 = 4, b^10 = 16, b^15 = 64 at scale 0.5), SURVEY.md's 5 710 032 entries.
LEVELS = {
    (0.5, "float32"): dict(resolution=[16, 22, 28, 37, 49, 65, 85, 112, 148, 195, 257, 338, 446, 589, 777, 1025],
                           offset=[0, 4096, 14744, 36696, 87352, 205008, 479640, 1003928, 1528216, 2052504, 2576792, 3101080, 3625368, 4149656,
                                   4673944, 5198232, 5722520], n_params=11445040),
    (0.5, "exact"): dict(resolution=[16, 22, 28, 37, 49, 64, 85, 112, 148, 195, 256, 338, 446, 589, 777, 1024],
                         offset=[0, 4096, 14744, 36696, 87352, 205008, 467152, 991440, 1515728, 2040016, 2564304, 3088592, 3612880, 4137168,
                                 4661456, 5185744, 5710032], n_params=11420064),
    (16.0, "float32"): dict(resolution=[16, 27, 45, 74, 123, 204, 338, 562, 934, 1553, 2581, 4290, 7132, 11857, 19711, 32768],
                            offset=[0, 4096, 23784, 114912, 520136, 1044424, 1568712, 2093000, 2617288, 3141576, 3665864, 4190152, 4714440,
                                    5238728, 5763016, 6287304, 6811592], n_params=13623184),
    (16.0, "exact"): dict(resolution=[16, 27, 45, 74, 123, 204, 338, 562, 934, 1553, 2581, 4290, 7132, 11857, 19711, 32768],
                          offset=[0, 4096, 23784, 114912, 520136, 1044424, 1568712, 2093000, 2617288, 3141576, 3665864, 4190152, 4714440,
                                  5238728, 5763016, 6287304, 6811592], n_params=13623184),
}


def per_level_scale(scale):
    return math.exp(math.log(2048 * scale / 16) / 15)       # models/networks.py:33
