"""
Seeded synthetic inputs shared by the golden-vector generator (make_golden.py, run once in the build
container against the imported reference) and by the tests (which regenerate the same inputs and
compare against the stored reference outputs).  Pure NumPy, no reference import.
"""

import numpy as np


def coefficients(seed, max_degree, scale=1e-10):
    """anm [N+1, N+1] ~ N(0,1) * scale  (default_rng(k), N(0,1)*1e-10)."""
    return np.random.default_rng(seed).standard_normal((max_degree + 1, max_degree + 1)) * scale


def spd_covariance(seed, size, scale=1e-22):
    """Seeded SPD matrix (G G^T)/k * scale with k = size + 16."""
    k = size + 16
    G = np.random.default_rng(seed).standard_normal((size, k))
    return (G @ G.T) / k * scale


def orderwise_random_blocks(seed, nmax):
    """[order0_cos, order1_cos, order1_sin, ...] random dense blocks, block m has shape (nmax+1-m)^2."""
    rng = np.random.default_rng(seed)
    blocks = [rng.standard_normal((nmax + 1, nmax + 1)) / (nmax + 1)]
    for m in range(1, nmax + 1):
        blocks.append(rng.standard_normal((nmax + 1 - m, nmax + 1 - m)) / (nmax + 1))
        blocks.append(rng.standard_normal((nmax + 1 - m, nmax + 1 - m)) / (nmax + 1))
    return blocks


def orderwise_normal_blocks(seed, nmax, scale=1e12):
    """Synthetic SPD order-wise normal-equation blocks (G G^T) * scale, reference block shapes."""
    rng = np.random.default_rng(seed)
    sizes = [nmax + 1]
    for m in range(1, nmax + 1):
        sizes += [nmax + 1 - m, nmax + 1 - m]
    blocks = []
    for s in sizes:
        G = rng.standard_normal((s, s + 4))
        blocks.append((G @ G.T) * scale)
    return blocks


def scattered_points(seed, count):
    """Longitude/latitude [rad] of `count` scattered points."""
    rng = np.random.default_rng(seed)
    lon = rng.uniform(-np.pi, np.pi, count)
    lat = np.arcsin(rng.uniform(-1, 1, count))
    return lon, lat


SPECIAL_COLAT = np.array([1e-3, 0.5 * np.pi, np.pi - 1e-3])


def var_covariance_function(seed, dim, max_lag):
    """Covariance function [Sigma_0, ..., Sigma_max_lag] of a stable VAR(1) process x_t = Phi x_{t-1} + w_t
    (Sigma_k = Phi^k Sigma_0, Sigma_0 from the discrete Lyapunov equation): a valid multivariate covariance
    function of any dimension for the Yule-Walker construction of grates/lstsq.py:127-167."""
    import scipy.linalg as la
    rng = np.random.default_rng(seed)
    Phi = rng.standard_normal((dim, dim))
    Phi *= 0.6 / np.max(np.abs(np.linalg.eigvals(Phi)))
    G = rng.standard_normal((dim, dim + 4))
    Q = G @ G.T / dim
    S0 = la.solve_discrete_lyapunov(Phi, Q)
    S0 = 0.5 * (S0 + S0.T)
    out = [S0]
    for _ in range(max_lag):
        out.append(Phi @ out[-1])
    return out


def observation_normals(seed, epoch_count, dim, scale=1.0):
    """Per-epoch observation normal equations: (N_t [dim, dim] SPD, n_t [dim, 1], lPl_t, observation count)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(epoch_count):
        A = rng.standard_normal((dim + 5, dim))
        l = rng.standard_normal((dim + 5, 1))
        out.append((A.T @ A * scale, A.T @ l * scale, float(np.sum(l * l) * scale), dim + 5))
    return out


def gfc_file_text(seed, max_degree, with_header=True):
    """Synthetic ICGEM GFC file (bytes): header keys, comment lines, `gfc n m C S [sigmas]` records in degree order."""
    rng = np.random.default_rng(seed)
    lines = ['product_type gravity_field', 'modelname synthetic_{0}'.format(seed)]
    if with_header:
        lines += ['earth_gravity_constant 3.986004418000E+14', 'radius 6.378136460000E+06']
    lines += ['max_degree {0}'.format(max_degree), 'norm fully_normalized', 'tide_system zero_tide', 'errors formal',
              'key    L    M        C                  S              sigma C       sigma S', 'end_of_head ====']
    for n in range(max_degree + 1):
        for m in range(n + 1):
            c, s = rng.standard_normal(2) * 1e-6 / max(n, 1) ** 2
            if m == 0:
                s = 0.0
            lines.append('gfc {0:4d} {1:4d} {2: .12E} {3: .12E} {4:.4E} {5:.4E}'.format(n, m, c, s, abs(c) * 1e-3, abs(s) * 1e-3))
    return ('\n'.join(lines) + '\n').encode('ascii')


def gsm_file_text(seed, max_degree, start='2010-03-01T00:00:00.00', end='2010-03-31T23:59:59.99'):
    """Synthetic GRACE SDS level-2 file (bytes): YAML header with the keys the loaders read, `GRCOF2` records."""
    rng = np.random.default_rng(seed)
    header = '''header:
  dimensions:
    degree: {0}
    order: {0}
  global_attributes:
    title: synthetic GSM product
    time_coverage_start: {1}
    time_coverage_end: {2}
  non-standard_attributes:
    earth_gravity_param:
      long_name: gravitational constant times mass of Earth
      units: m3/s2
      value: 3.9860044150E+14
    mean_equator_radius:
      long_name: mean equator radius
      units: meters
      value: 6.3781363000E+06
# End of YAML header
'''.format(max_degree, start, end)
    lines = []
    for n in range(max_degree + 1):
        for m in range(n + 1):
            c, s = rng.standard_normal(2) * 1e-6 / max(n, 1) ** 2
            if m == 0:
                s = 0.0
            lines.append('GRCOF2 {0:4d} {1:4d} {2: .12E} {3: .12E} {4:.4E} {5:.4E} 20100301.0000 20100401.0000 nnnn'.format(n, m, c, s, abs(c) * 1e-3, abs(s) * 1e-3))
    return (header + '\n'.join(lines) + '\n').encode('ascii')


def sinex_file_text(seed, min_degree, max_degree, lower=False):
    """Synthetic SINEX normal-equation file (bytes), storage scheme 6c: statistics, right-hand side vector with one record
    per coefficient (degree-wise order, columns as in the SINEX 2.02 parameter records) and the normal matrix as one
    triangle in rows of up to three values."""
    rng = np.random.default_rng(seed)
    names = []
    for n in range(min_degree, max_degree + 1):
        names.append(('CN', n, 0))
        for m in range(1, n + 1):
            names.append(('CN', n, m))
            names.append(('SN', n, m))
    p = len(names)
    g = rng.standard_normal((p + 8, p))
    normals = g.T @ g
    rhs = normals @ (rng.standard_normal(p) * 1e-9)
    out = ['%=SNX 2.02 TST 26:001:00000 TST 02:091:00000 02:120:86399 C {0:5d} 2'.format(p),
           '*-------------------------------------------------------------------------------',
           '+FILE/REFERENCE',
           ' DESCRIPTION        synthetic normal equations',
           ' OUTPUT             parity fixture',
           '-FILE/REFERENCE',
           '+SOLUTION/STATISTICS',
           '*_STATISTICAL PARAMETER________ __VALUE(S)____________']
    for label, value in (('NUMBER OF OBSERVATIONS', 12345.0), ('NUMBER OF UNKNOWNS', float(p)), ('NUMBER OF DEGREES OF FREEDOM', 12345.0 - p),
                         ('WEIGHTED SQUARE SUM OF O-C', 9.87654321012345e+03)):
        out.append(' {0:<30s} {1:22.15e}'.format(label, value))
    out += ['-SOLUTION/STATISTICS', '+SOLUTION/NORMAL_EQUATION_VECTOR', '*INDEX TYPE__ CODE PT SOLN _REF_EPOCH__ UNIT S __RIGHT_HAND_SIDE____']
    for k, (cs, n, m) in enumerate(names):
        out.append(' {0:5d} {1:<6s} {2:4d} -- {3:4d} 02:106:00000 ---- 2 {4:21.14e}'.format(k + 1, cs, n, m, rhs[k]))
    out.append('-SOLUTION/NORMAL_EQUATION_VECTOR')
    tag = 'SOLUTION/NORMAL_EQUATION_MATRIX ' + ('L' if lower else 'U')
    out += ['+' + tag, '*PARA1 PARA2 ____PARA2+0__________ ____PARA2+1__________ ____PARA2+2__________']
    for r in range(p):
        first, last = (0, r + 1) if lower else (r, p)
        for c in range(first, last, 3):
            out.append(' {0:5d} {1:5d}'.format(r + 1, c + 1) + ''.join(' {0:21.14e}'.format(v) for v in normals[r, c:min(c + 3, last)]))
    out += ['-' + tag, '%ENDSNX']
    return ('\n'.join(out) + '\n').encode('ascii')


def basis_function_case(seed, count, min_degree, max_degree):
    """Nodal points, values and shape factors of the space-domain representations: K_rbf [N+1, N+1] (coefficient layout),
    K_aniso [P, P] for the band min_degree .. max_degree."""
    rng = np.random.default_rng(seed)
    lon, lat = scattered_points(seed + 1, count)
    values = rng.standard_normal(count) * 1e-3
    k_rbf = rng.uniform(0.5, 1.5, (max_degree + 1, max_degree + 1))
    p = (max_degree + 1) ** 2 - min_degree ** 2
    k_aniso = rng.standard_normal((p, p)) / p
    return lon, lat, values, k_rbf, k_aniso


def api_probes(pkg):
    """(label, thunk) pairs that exercise the error contract of the path's public API on `pkg` -- the reference package in
    make_golden.py (which records the exception TYPE each one raises in g19_api.json), grates_amd in tests/test_api_signatures.py.  Every
    thunk is cheap, host-only and needs no data file (the DDK normal blocks are absent from the reference mount)."""
    gf = pkg.gravityfield.PotentialCoefficients()
    gf.anm = coefficients(5, 6)
    small = pkg.gravityfield.PotentialCoefficients()
    small.anm = coefficients(6, 4)
    grid = pkg.grid.GeographicGrid(30.0, 30.0)
    blocks4 = orderwise_random_blocks(7, 4)
    ewh = pkg.kernel.get_kernel('ewh')

    def set_values(target, val):
        target.values = val

    return [
        ('pc_add_str', lambda: gf + 'x'),
        ('pc_sub_int', lambda: gf - 1),
        ('pc_mul_pc', lambda: gf * gf),
        ('pc_div_str', lambda: gf / 'x'),
        ('pc_add_pc', lambda: gf + small),
        ('pc_mul_float', lambda: gf * 2.0),
        ('pc_values_2d', lambda: set_values(gf.copy(), np.zeros((7, 7)))),
        ('pc_values_list', lambda: set_values(gf.copy(), [1.0, 2.0])),
        ('pc_values_none_ok', lambda: set_values(gf.copy(), None)),
        ('grid_values_2d', lambda: set_values(grid.copy(), np.zeros((6, 12)))),
        ('grid_values_size', lambda: set_values(grid.copy(), np.zeros(5))),
        ('grid_values_list', lambda: set_values(grid.copy(), [0.0] * 72)),
        ('grid_values_ok', lambda: set_values(grid.copy(), np.zeros(72))),
        ('grid_analysis_without_values', lambda: grid.copy().to_potential_coefficients(0, 2)),
        ('get_kernel_unknown', lambda: pkg.kernel.get_kernel('no_such_kernel')),
        ('get_kernel_geoid', lambda: pkg.kernel.get_kernel('geoid')),
        ('kernel_shape_mismatch', lambda: ewh.inverse_coefficients(0, 4, np.full(3, 6.378e6), np.full(4, 1.0))),
        ('kernel_list_input', lambda: ewh.inverse_coefficients(0, 4, [6.378e6], [1.0])),
        ('kernel_scalar_ok', lambda: ewh.inverse_coefficients(0, 4, 6.378e6, 1.0)),
        ('gaussian_filter_non_pc', lambda: pkg.filter.Gaussian(300).filter(5)),
        ('orderwise_filter_non_pc', lambda: pkg.filter.OrderWiseFilter(blocks4).filter(np.zeros((5, 5)))),
        ('orderwise_filter_degree_too_high', lambda: pkg.filter.OrderWiseFilter(blocks4).filter(gf)),
        ('orderwise_filter_ok', lambda: pkg.filter.OrderWiseFilter(blocks4).filter(small)),
        ('ddk_generic_level_0', lambda: pkg.filter.DDKGeneric(0)),
        ('general_matrix_not_square', lambda: pkg.filter.GeneralMatrix(np.zeros((3, 4)), 0, 1)),
        ('general_matrix_wrong_size', lambda: pkg.filter.GeneralMatrix(np.zeros((5, 5)), 0, 1)),
        ('general_matrix_ok', lambda: pkg.filter.GeneralMatrix(np.eye(4), 0, 1)),
        ('unravel_wrong_length', lambda: pkg.utilities.unravel_coefficients(np.zeros(7))),
        ('ravel_ok', lambda: pkg.utilities.ravel_coefficients(np.zeros((4, 4)), 1, 3)),
        ('legendre_scalar_colat', lambda: pkg.utilities.legendre_functions(3, np.array([0.4]))),
    ]
