"""The condition under which the TIES kernels of csrc/raster_mesh.hip mark a pixel for the replay of the reference's CUDA
procedure (`eval_candidates`, DESIGN.md section 4 "Exact depth ties and the CUDA tie order"), checked by brute force.

The reference's CUDA kernels keep a pixel's K nearest faces in an UNSORTED array filled in ascending face index: while the array
has room a hit is appended; once it is full a hit replaces "the" largest entry iff it is STRICTLY nearer, and "the" largest entry
is then found by a scan with a strict comparison that starts from the newcomer (rasterize_meshes.cu:216-237).  Our kernels keep
the K smallest under the total order (z, face index).  Claim: with zK = the K-th smallest depth, S = the hits nearer than zK,
T = the hits at zK, m = K - |S|, the two survivor sets can differ only if T has more than m members AND some member of S has a
larger face index than the (m + 1)-th member of T -- the smallest index the total order drops.  The kernels mark exactly those
pixels (plus every pixel the clipped-face neighbour rule touched); all others keep the fine kernel's output.
"""
import random


def total_order(hits, K):
    return sorted(hits)[:K]


def cuda_procedure(hits, K):
    """hits: (z, idx) in ascending idx.  Returns the survivors sorted by (z, idx) (rasterize_meshes.cu:30-32)."""
    q = []
    qmax_z, qmax_i = -1000.0, -1
    for z, f in hits:
        if len(q) < K:
            q.append((z, f))
            if z > qmax_z:
                qmax_z, qmax_i = z, len(q) - 1
        elif z < qmax_z:
            q[qmax_i] = (z, f)
            qmax_z = z
            for j in range(K):
                if q[j][0] > qmax_z:
                    qmax_z, qmax_i = q[j][0], j
    return sorted(q)


def marked(hits, K):
    if len(hits) <= K or K <= 1:
        return False
    zK = sorted(h[0] for h in hits)[K - 1]
    S = [f for z, f in hits if z < zK]
    T = sorted(f for z, f in hits if z == zK)
    m = K - len(S)
    dropped = T[m:]
    return bool(dropped) and bool(S) and max(S) > dropped[0]


def test_pixels_that_are_not_marked_have_the_references_survivors():
    rng = random.Random(5)
    differ = marked_and_equal = unmarked = 0
    for _ in range(60000):
        n = rng.randint(1, 14)
        K = rng.randint(1, 8)
        levels = rng.randint(1, 5)  # few depth levels: ties everywhere
        hits = [(float(rng.randint(0, levels)), f) for f in sorted(rng.sample(range(40), n))]
        ours, theirs = total_order(hits, K), cuda_procedure(hits, K)
        assert [z for z, _ in ours] == [z for z, _ in theirs]  # the same depths either way
        if marked(hits, K):
            differ += ours != theirs
            marked_and_equal += ours == theirs
        else:
            unmarked += 1
            assert ours == theirs, (hits, K, ours, theirs)
    assert differ > 1000 and unmarked > 1000  # the generator exercises both sides
    print(f"marked and different {differ}, marked but equal {marked_and_equal} (the mark is conservative), not marked {unmarked}")


def test_one_entry_queues_never_differ():
    rng = random.Random(6)
    for _ in range(5000):
        n = rng.randint(1, 10)
        hits = [(float(rng.randint(0, 2)), f) for f in sorted(rng.sample(range(30), n))]
        assert total_order(hits, 1) == cuda_procedure(hits, 1)
