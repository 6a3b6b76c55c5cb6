"""numpy model of the GPU work decomposition (pure numpy; no oracle, no reference).

Each source is split into 4 quadrants x 2 octants = 8 independent work units (one
wavefront each on the GPU) plus a zero-fill unit:

  X-octant (|dx| > |dy|): lane <-> row offset j, marches along x (step = i).
      new[j] = (a - c*(a - b)) * occ(i, j),  a = prev[j], b = prev[j-1], c = j/i
      and the diagonal cell of step i is new[i] = new[i-1] * occ(i, i)   (quirk Q1)
  Y-octant (|dy| > |dx|): lane <-> column offset i, marches along y (step = j).
      new[i] = (a - c*(a - b)) * occ(i, j),  a = prev[i], b = prev[i-1], c = i/j
      and lane j is seeded with diag(j), which the unit recomputes itself from the
      two-term recurrence sub(j) = V(j, j-1), diag(j) = sub(j) * occ(j, j).

Every step depends only on the previous step of the same unit, so a step is
embarrassingly parallel across lanes: that is the "front" the kernels sweep.
tests/test_schedule_model.py checks this model bit-for-bit against the oracle.
"""
import numpy as np

QUADS = ((+1, +1), (-1, +1), (-1, -1), (+1, -1))  # Q1..Q4, reference order


def quadrant_extents(nx, ny, sx, sy, dirx, diry):
    # negative directions stop one cell short of the border (quirk Q2)
    ni = nx - sx if dirx > 0 else sx
    nj = ny - sy if diry > 0 else sy
    return ni, nj


def _stencil(a, b, c):
    return a - c * (a - b)


def sweep_units(occ, sx, sy):
    """Returns the field produced by the 8 units + zero fill, float64 [ny, nx]."""
    ny, nx = occ.shape
    f = occ.astype(np.float64)
    out = np.full((ny, nx), np.nan)
    # zero-fill unit: row 0 / column 0 when no quadrant covers them
    if sx > 0:
        out[:, 0] = 0.0
    if sy > 0:
        out[0, :] = 0.0
    for dirx, diry in QUADS:
        ni, nj = quadrant_extents(nx, ny, sx, sy, dirx, diry)
        if ni <= 0 or nj <= 0:
            continue
        X = lambda i: sx + dirx * i
        Y = lambda j: sy + diry * j
        origin = 1.0 * f[sy, sx]
        # ---- X-octant unit -------------------------------------------------
        prev = np.zeros(nj + 1)
        prev[0] = origin
        out[sy, sx] = origin
        for i in range(1, ni):
            m = min(i, nj)  # active lanes j < m
            j = np.arange(m)
            a = prev[:m]
            b = np.concatenate(([0.0], prev[: m - 1]))
            c = j / float(i)
            new = _stencil(a, b, c) * f[Y(j), X(i)]
            out[Y(j), X(i)] = new
            prev[:m] = new
            if i < nj:  # diagonal cell: inherits the cell just "below" it (Q1)
                d = new[i - 1] * f[Y(i), X(i)]
                prev[i] = d
                out[Y(i), X(i)] = d
        # ---- Y-octant unit -------------------------------------------------
        prev = np.zeros(ni + 1)
        prev[0] = origin
        diag, sub = origin, 0.0
        for j in range(1, nj):
            m = min(j, ni)  # active lanes i < m
            i = np.arange(m)
            a = prev[:m]
            b = np.concatenate(([0.0], prev[: m - 1]))
            c = i / float(j)
            new = _stencil(a, b, c) * f[Y(j), X(i)]
            out[Y(j), X(i)] = new
            prev[:m] = new
            if j < ni:  # seed lane j with diag(j) from the private 1-D recurrence
                sub = _stencil(diag, sub, (j - 1) / float(j)) * f[Y(j - 1), X(j)]
                diag = sub * f[Y(j), X(j)]
                prev[j] = diag
    return out


def first_touch_rank(nx, ny, sx, sy, x, y):
    """Push-order rank of cell (x, y) in updateVisibility (quirk Q6); lower = pushed earlier."""
    dx, dy = x - sx, y - sy
    if dx >= 0 and dy >= 0:
        q, r = 0, dx * (ny - sy) + dy
    elif dx < 0 and dy >= 0:
        q, r = 1, (-dx) * (ny - sy) + dy
    elif dy < 0 and (dx < 0 or (dx == 0 and sx >= 1)):
        q, r = 2, (-dx) * sy + (-dy)
    else:
        q, r = 3, dx * sy + (-dy)
    return (q << 40) | r
