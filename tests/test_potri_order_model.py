"""CPU model checks of the item lists of the single-launch factorisation + inverse (csrc/kernels_chol.hip), restated in Python:

* potri_team deals the items of the fused inverse -- T(j), P(i), X(i, j), K(i, j) -- round-robin in ONE global order.  A worker runs
  the first ready item of its list, so the launch cannot deadlock if (and this is what is checked) every input of an item is
  produced by an item EARLIER in that order (or by the factorisation, which never waits for the inverse): the globally first
  unfinished item is then always the first unfinished item of its owner, and its inputs are finished.
* the workers of the factorisation own the lower tiles; the tiles (i, k) with 1 <= i - k <= band have one owner per 64-column half.
  The enumeration the kernel decodes (column by column: the nb - k tiles, then the second halves) must hit every (tile, half)
  exactly once and agree with the host's item count.

No GPU, no library: these are the invariants the kernel's deal loops rely on; the GPU suite checks the results."""
import itertools

import pytest


def inverse_items(nb, plast):
    """The global order of potri_team's deal loop: per row r: T(r), P(r) (plast, r > 0), X(r, 0..r-1); then K(r, 0..r) by row."""
    order = []
    for r in range(nb):
        order.append(("T", r, r))
        if r > 0 and plast:
            order.append(("P", r, r - 1))
        for j in range(r):
            order.append(("X", r, j))
    for r in range(nb):
        for j in range(r + 1):
            order.append(("K", r, j))
    return order


def inverse_inputs(item, nb, plast):
    """Items of the inverse an item waits for (potri_item_ready); inputs from the factorisation (factored[], panel_done[]) are not
    listed: the factorisation never waits for the inverse."""
    kind, i, j = item
    if kind == "T":
        return []
    if kind == "P":                     # T(i) and the panel tile (i, i-1)
        return [("T", i, i)]
    if kind == "X":
        nterms = i - j - (1 if plast else 0)
        deps = [("T", k, k) if k == j else ("X", k, j) for k in range(j, j + nterms)]     # U_jk for the accumulated terms
        deps.append(("T", i, i))                                                            # Q = M T_ii^T
        if plast:
            deps.append(("P", i, i - 1))
            deps.append(("T", j, j) if i - 1 == j else ("X", i - 1, j))                     # U_{j,i-1} for the last step
        return deps
    deps = []                                                                               # K(i, j): U_ik, U_jk for k >= i
    for k in range(i, nb):
        deps.append(("T", i, i) if k == i else ("X", k, i))
        deps.append(("T", j, j) if k == j else ("X", k, j))
    return deps


@pytest.mark.parametrize("nb", [3, 4, 7, 12, 16, 24, 32])
@pytest.mark.parametrize("plast", [0, 1])
def test_inverse_items_only_wait_for_earlier_items(nb, plast):
    order = inverse_items(nb, plast)
    pos = {it: n for n, it in enumerate(order)}
    assert len(pos) == len(order) == nb + (nb - 1) * plast + nb * (nb - 1) // 2 + nb * (nb + 1) // 2
    for it in order:
        for dep in inverse_inputs(it, nb, plast):
            assert dep in pos, (it, dep)
            assert pos[dep] < pos[it], (it, dep)
    # the round-robin deal keeps every worker's list in the global order (the kernel scans a list front to back)
    for G2 in (8, 57, 158):
        for w in range(min(G2, 4)):
            mine = [n for n in range(len(order)) if n % G2 == w]
            assert mine == sorted(mine)


def factor_items(nb, band):
    """The enumeration of the workers' deal loop: column k carries its nb - k tiles ((k, k) first) and then the second halves of the
    tiles (k+1, k) .. (k+band, k); item 0 -- tile (0, 0), the chain's -- is skipped.  Returns (i, k, half) with half 0 = whole tile,
    1 / 2 = columns 0-63 / 64-127."""
    items = []
    for k in range(nb):
        cnt = nb - k + min(band, nb - 1 - k)
        for e in range(cnt):
            second = e >= nb - k
            i = k + 1 + (e - (nb - k)) if second else k + e
            half = 2 if second else (1 if 1 <= e <= band else 0)
            items.append((i, k, half))
    return items[1:]


@pytest.mark.parametrize("nb,band", [(4, 0), (4, 1), (4, 3), (16, 1), (16, 15), (32, 1), (32, 4), (40, 39)])
def test_every_tile_half_has_exactly_one_owner(nb, band):
    items = factor_items(nb, band)
    n_second = sum(min(band, nb - 1 - k) for k in range(nb))
    assert len(items) == nb * (nb + 1) // 2 - 1 + n_second              # the host's `tiles` (launch_potrf_dataflow_impl)
    assert len(set(items)) == len(items)
    owned = {}
    for i, k, half in items:
        assert 0 <= k <= i < nb
        owned.setdefault((i, k), []).append(half)
    for i, k in itertools.product(range(nb), repeat=2):
        if k > i or (i, k) == (0, 0):
            assert (i, k) not in owned
        elif 1 <= i - k <= band:
            assert sorted(owned[(i, k)]) == [1, 2], (i, k, owned[(i, k)])
        else:
            assert owned[(i, k)] == [0], (i, k, owned[(i, k)])
    # dealt t = widx, widx + W, ...: a worker's items come in increasing column order (the column is what is needed first)
    for W in (7, 96, 254):
        for w in range(min(W, 3)):
            cols = [items[t][1] for t in range(w, len(items), W)]
            assert cols == sorted(cols)
