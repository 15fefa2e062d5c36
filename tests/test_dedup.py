"""a18 / f2: de-duplication (predict_wsi.py:896-965).  Default = ``geojson.dedup_exact``: scipy's own pair set walked in
ITS order, result-identical to the reference (equality tests below).  Opt-in fast path = the device radius search
(cpx_dedup_pairs, exact pair set) + ``geojson.dedup_from_pairs`` (models the set order; documented tolerance inside
clusters of >= 3 cells only)."""
import numpy as np
import pytest
from scipy.spatial import KDTree

from classpose_amd import geojson


def _clustered(rng, n_cells, extent, tie_frac=0.3):
    """centroids as the tile loop produces them: most cells once, some 2x (edge overlaps), some 4x (corner
    overlaps), a few chains; a share of the copies with EQUAL areas (flow-injection slides), 2-decimal rounding"""
    base = rng.uniform(0, extent, (n_cells, 2))
    area = rng.integers(30, 400, n_cells).astype(np.float64)
    mult = rng.choice([1, 2, 3, 4], n_cells, p=[0.6, 0.25, 0.05, 0.10])
    pts, sz = [], []
    for b, a, m in zip(base, area, mult):
        for k in range(m):
            pts.append(b + rng.uniform(-2.5, 2.5, 2) * (k > 0))
            sz.append(a if rng.random() < tie_frac else a + rng.integers(-5, 6))
    chain = np.array([[extent + 50 + 5.0 * k, 50.0] for k in range(7)])            # A-B-C-... chain, 5 px apart
    pts += chain.tolist(); sz += [50, 60, 55, 60, 10, 70, 70]
    order = rng.permutation(len(pts))
    return np.round(np.asarray(pts)[order], 2), np.asarray(sz, np.float64)[order]


def _scipy_pairs(c, r=7.5):
    p = KDTree(c).query_pairs(r, output_type="ndarray").astype(np.int32)
    return p[np.lexsort((p[:, 1], p[:, 0]))] if len(p) else p.reshape(0, 2)


def test_dedup_from_pairs_equals_reference_loop_golden(golden):
    """kept ids of the reference's own deduplicate (golden fixture incl. the A-B-C chain)"""
    _, js = golden
    pts = np.array(js["geojson"]["points"])
    pairs = _scipy_pairs(pts[:, :2])
    assert geojson.dedup_from_pairs(len(pts), pts[:, 2], pairs).tolist() == js["geojson"]["kept_ids"]


def test_dedup_exact_equals_reference_golden(golden):
    """the default path against the kept ids of the reference's own deduplicate (incl. the A-B-C chain)"""
    _, js = golden
    pts = np.array(js["geojson"]["points"])
    st = {}
    assert geojson.dedup_exact(pts[:, :2], pts[:, 2], stats=st).tolist() == js["geojson"]["kept_ids"]
    assert st["n_pairs"] == len(_scipy_pairs(pts[:, :2])) and st["n_order_dependent"] >= 3


@pytest.mark.parametrize("seed,n_cells", [(0, 3000), (1, 3000), (2, 3000), (3, 60000)])
def test_dedup_exact_equals_verbatim_loop(seed, n_cells):
    """EQUALITY (no tolerance) of the default path with the reference's loop run verbatim over scipy's own set
    (geojson.dedup_indices, golden-pinned), on cell tables full of >= 3-cell clusters, equal areas and chains"""
    c, a = _clustered(np.random.default_rng(seed), n_cells, 4000.0 * (n_cells / 3000) ** 0.5)
    st = {}
    got = geojson.dedup_exact(c, a, stats=st).tolist()
    ref = geojson.dedup_indices(c.tolist(), a.tolist())
    assert got == ref
    assert st["n_order_dependent"] > 0.1 * n_cells                       # the order-dependent case is what is being tested
    assert st["n_order_dependent"] == geojson.count_order_dependent(len(c), _scipy_pairs(c))


def test_dedup_exact_trivial_inputs():
    assert geojson.dedup_exact(np.zeros((0, 2)), np.zeros(0)).tolist() == []
    assert geojson.dedup_exact(np.array([[0.0, 0.0], [100.0, 0.0]]), np.ones(2)).tolist() == [0, 1]
    assert geojson.dedup_exact(np.array([[0.0, 0.0], [3.0, 0.0]]), np.array([5.0, 5.0])).tolist() == [0]      # tie keeps i
    assert geojson.dedup_exact(np.array([[0.0, 0.0], [3.0, 0.0]]), np.array([5.0, 6.0])).tolist() == [1]


def _reference_loop(n, sizes, neighbours):
    """predict_wsi.py:929-960 verbatim over a given pair set"""
    groups, member_to_group = {}, {}
    for pair in neighbours:
        if (pair[0] not in member_to_group) and (pair[1] not in member_to_group):
            group_idx = len(groups)
            groups[group_idx] = []
            member_to_group[pair[0]] = group_idx
            member_to_group[pair[1]] = group_idx
        else:
            group_idx = member_to_group[pair[0]] if pair[0] in member_to_group else member_to_group[pair[1]]
        if pair[0] not in groups[group_idx]:
            groups[group_idx].append(pair[0])
        if pair[1] not in groups[group_idx]:
            groups[group_idx].append(pair[1])
    to_remove = {}
    for k in groups:
        group = groups[k]
        if len(group) > 1:
            largest = group[np.argmax([sizes[i] for i in group])]
            for i in group:
                if i != largest and i not in to_remove:
                    to_remove[i] = True
    return [i for i in range(n) if i not in to_remove]


def _assert_same_up_to_ambiguous_clusters(n, pairs, got, ref, max_frac):
    """kept ids may differ only inside connected components of >= 3 cells (where the reference's own outcome
    depends on hash-collision displacement inside its Python set), and only in a small fraction of them"""
    diff = set(ref) ^ set(got)
    assert len(diff) <= max_frac * n, (len(diff), n)
    deg = np.bincount(pairs.ravel(), minlength=n)
    nb = {i: set() for i in diff}
    for i, j in pairs[np.isin(pairs[:, 0], list(diff)) | np.isin(pairs[:, 1], list(diff))].tolist():
        if i in nb: nb[i].add(j)
        if j in nb: nb[j].add(i)
    assert all(deg[i] >= 2 or any(deg[j] >= 2 for j in nb[i]) for i in diff)    # never in a 2-cell component


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_dedup_from_pairs_equals_reference_loop_random(seed):
    """the OPT-IN fast path (vectorised 2-cell components + the reference's loop over the remaining pairs in hash-slot
    order; CLASSPOSE_DEDUP_BACKEND=device) against (1) the reference's loop run verbatim over a Python set of the same pairs and (2) the loop
    over scipy's own set (geojson.dedup_indices, golden-pinned): identical outside >= 3-cell clusters, and
    inside them up to the collision-displacement ambiguity of the reference's set order, ties and chains included"""
    c, a = _clustered(np.random.default_rng(seed), 3000, 4000.0)
    pairs = _scipy_pairs(c)
    got = geojson.dedup_from_pairs(len(c), a, pairs).tolist()
    same_set = set(zip(pairs[:, 0].tolist(), pairs[:, 1].tolist()))
    _assert_same_up_to_ambiguous_clusters(len(c), pairs, got, _reference_loop(len(c), a.tolist(), same_set), 0.002)
    ref = geojson.dedup_indices(c.tolist(), a.tolist())
    assert 0.55 * len(c) < len(ref) < 0.8 * len(c)
    _assert_same_up_to_ambiguous_clusters(len(c), pairs, got, ref, 0.002)


def test_set_order_model_matches_cpython():
    """hash((i, j)) and the table mask are CPython's; slot order == real set iteration order up to collisions"""
    rng = np.random.default_rng(0)
    a = rng.integers(0, 3_000_000, 500)
    b = a + rng.integers(1, 5000, 500)
    h = geojson._tuple2_hash(a, b)
    assert all(int(x) == (hash((int(i), int(j))) & 0xFFFFFFFFFFFFFFFF) for x, i, j in zip(h, a, b))
    for n in (1, 5, 6, 19, 20, 77, 307, 1229, 3120, 50001, 120000):
        assert geojson._fast_set_table_mask(n) == geojson._set_table_mask(n), n
    pairs = np.unique(np.stack([a, b], 1), axis=0)
    real = list(set(zip(pairs[:, 0].tolist(), pairs[:, 1].tolist())))
    slot = geojson._tuple2_hash(pairs[:, 0], pairs[:, 1]) & np.uint64(geojson._fast_set_table_mask(len(pairs)))
    pred = [tuple(x) for x in pairs[np.argsort(slot, kind="stable")].tolist()]
    pos = {t: i for i, t in enumerate(real)}
    assert sum(pos[x] > pos[y] for x, y in zip(pred, pred[1:])) <= 0.08 * len(pred)


def test_dedup_from_pairs_no_pairs_and_empty():
    assert geojson.dedup_from_pairs(5, np.ones(5), np.zeros((0, 2), np.int32)).tolist() == [0, 1, 2, 3, 4]
    assert geojson.dedup_from_pairs(0, np.ones(0), np.zeros((0, 2), np.int32)).tolist() == []


@pytest.mark.gpu
def test_device_pairs_equal_kdtree_query_pairs_golden(cuda, golden):
    from classpose_amd import ops
    _, js = golden
    pts = np.array(js["geojson"]["points"])
    pairs = ops.dedup_pairs(pts[:, :2], 7.5, cuda)
    assert np.array_equal(pairs, _scipy_pairs(pts[:, :2]))                    # incl. the 7.4 / 7.6 px boundary cases
    assert geojson.dedup_from_pairs(len(pts), pts[:, 2], pairs).tolist() == js["geojson"]["kept_ids"]


@pytest.mark.gpu
@pytest.mark.parametrize("n_cells,extent", [(500, 700.0), (650_000, 40_000.0)])
def test_device_pairs_equal_kdtree_query_pairs_synthetic(cuda, n_cells, extent):
    """~10^6 centroids spread like a 40k^2 slide's cell table: identical pair SET, then identical kept ids"""
    import time
    from classpose_amd import ops
    c, a = _clustered(np.random.default_rng(7), n_cells, extent)
    # exact-radius cases: partners at distance exactly 7.5 (3-4-5 triangle scaled) must be included (<=)
    c[:2] = [[100.0, 100.0], [104.5, 106.0]]
    ops.dedup_pairs(c[:1000], 7.5, cuda)                                      # warm-up
    t0 = time.perf_counter()
    pairs = ops.dedup_pairs(c, 7.5, cuda)
    t1 = time.perf_counter()
    keep = geojson.dedup_from_pairs(len(c), a, pairs)
    t2 = time.perf_counter()
    ref = _scipy_pairs(c)
    assert np.array_equal(pairs, ref)
    assert (0, 1) in set(map(tuple, pairs[:50].tolist()))
    print(f"{len(c)} centroids, {len(pairs)} pairs: device search {1e3 * (t1 - t0):.1f} ms incl. H2D/D2H, "
          f"host grouping {1e3 * (t2 - t1):.1f} ms")
    t3 = time.perf_counter()
    full = _reference_loop(len(c), a.tolist(), set(zip(pairs[:, 0].tolist(), pairs[:, 1].tolist())))
    print(f"   (the reference's loop over the whole pair set: {time.perf_counter() - t3:.2f} s)")
    _assert_same_up_to_ambiguous_clusters(len(c), pairs, keep.tolist(), full, 0.002)
