"""CPU: the masked object-level walk (DESIGN.md §5 "masked FRONT", kernels.hip front_masked) against the LITERAL walk RS:294-326.
The kernel never walks the heap: it evaluates one slab-test bit per node and derives, with shifts and masks over the nodes in POP
order, which nodes the reference's stack walk would pop (its BVHNode fetch count) and which MeshObjects it would test, in which order
(`tests` is never reset: A.5).  Here the table the library builds (urt_debug_build_walk_table) is interpreted in numpy exactly as the
kernel does, for random heaps — complete and truncated arrays, leaves high up, fillers, empty boxes, out-of-range ids, empty meshes —
and for EVERY hit pattern a ray could produce on them (or 4,096 random ones), and compared with a literal simulation of the shader."""
import ctypes as C

import numpy as np
import pytest

from unityraytracer_amd import _lib, scenes

EMPTY_ROOT = 0x7FFFFFFF


def build_table(heap, mesh_root, small_first=None):
    lib = _lib.load()
    out = np.zeros(4 * (20 + 2 * 32), np.float32)
    n = C.c_int()
    roots = np.ascontiguousarray(mesh_root, np.int32)
    sf = np.ascontiguousarray(small_first if small_first is not None else -np.ones(len(roots)), np.int32)
    rc = lib.urt_debug_build_walk_table(heap.ctypes.data_as(C.c_void_p) if len(heap) else None, len(heap), len(roots), roots.ctypes.data_as(C.c_void_p),
                                        sf.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
    assert rc == 0
    return out[:n.value].view(np.int32).copy(), out[:n.value].copy()


def literal_walk(heap, hit, n_meshes, mesh_root):
    """RS:294-326 with the net semantics of SURVEY A.5: returns (BVHNode fetches, [object ids tested, in order])."""
    stack, fetched, tested, seen = [0], 0, [], False
    while stack:
        bi = stack.pop()
        h, index = False, -1
        if bi < len(heap):
            fetched += 1
            index = int(heap[bi]["index"])
            nonempty = not np.array_equal(heap[bi]["vmin"], heap[bi]["vmax"])
            h = bool(hit[bi]) and nonempty                         # RS:273: an empty node never passes
        if h:
            if index < 0:
                stack.append(2 * bi + 1); stack.append(2 * bi + 2)  # the right child is popped first
            else:
                seen = True
        if seen and 0 <= index < n_meshes and mesh_root[index] != EMPTY_ROOT:
            tested.append(index)
    return fetched, tested


def masked_walk(words, hit_by_heap_index, heap):
    """The kernel's arithmetic on the table (front_masked)."""
    n_eval, levels, imask, exist = (int(words[k]) & 0xFFFFFFFF for k in range(4))
    leaf_any, leaf_valid = int(words[4]) & 0xFFFFFFFF, int(words[5]) & 0xFFFFFFFF
    dm = [int(words[8 + d]) & 0xFFFFFFFF for d in range(4)]
    ls = [int(words[12 + d]) for d in range(4)]
    pos_tab = words[16:80].reshape(32, 2)
    ev = words[80:80 + 8 * n_eval].reshape(n_eval, 8)
    # which heap node sits at which position: recover from the boxes is fragile; use the pop order of the complete tree instead
    N = (1 << levels) - 1
    order, st = [], [0]
    while st:
        i = st.pop(); order.append(i)
        if 2 * i + 2 < N:
            st.append(2 * i + 1); st.append(2 * i + 2)
    pos_of = {i: p for p, i in enumerate(order)}
    H = 0
    for e in range(n_eval):
        pbit, bit = int(ev[e, 3]) & 0xFFFFFFFF, int(ev[e, 7]) & 0xFFFFFFFF
        p = bit.bit_length() - 1
        if pbit and not (H & pbit):
            continue                                                # the wave-uniform skip (here: one ray)
        if hit_by_heap_index[order[p]]:
            H |= bit
    P = 1
    for d in range(4):
        if d + 1 >= levels:
            break
        X = P & H & imask & dm[d]
        P |= ((X << 1) | (X << ls[d])) & 0xFFFFFFFF
    fetched = bin(P & exist).count("1")
    src = P & H & leaf_any
    T = 0
    if src:
        first = (src & -src).bit_length() - 1
        T = P & leaf_valid & ~((1 << first) - 1)
    tested = []
    while T:
        p = (T & -T).bit_length() - 1
        T &= T - 1
        tested.append(p)
    return fetched, tested, pos_tab, pos_of


def random_heap(rng, n_objects, truncate):
    depth = max(1, int(np.ceil(np.log2(max(1, n_objects)))) + 1)
    n = (1 << depth) - 1
    heap = np.zeros(n, scenes.BVHNODE_DT)
    heap["index"] = -1
    # random shape: walk from the root, each node becomes a leaf with some probability (leaves high up), else interior
    ids = list(rng.permutation(n_objects + 2) - 1)                  # includes -1 ... n_objects: some ids out of range / negative-as-interior
    for i in range(n):
        lo = rng.uniform(-5, 5, 3).astype(np.float32)
        hi = lo + rng.uniform(0.1, 3, 3).astype(np.float32)
        kind = rng.random()
        last_level = 2 * i + 1 >= n
        if kind < 0.12:
            heap[i]["vmin"] = heap[i]["vmax"] = 0                   # a filler (RM:490-494)
        elif kind < 0.18:
            heap[i]["vmin"] = heap[i]["vmax"] = lo                  # empty bounds on a would-be interior or leaf node
            heap[i]["index"] = int(rng.integers(-1, n_objects))
        else:
            heap[i]["vmin"], heap[i]["vmax"] = lo, hi
            if last_level or kind > 0.62:
                heap[i]["index"] = int(ids[int(rng.integers(0, len(ids)))]) if rng.random() < 0.85 else n_objects + int(rng.integers(0, 50))
    if truncate:
        heap = heap[: int(rng.integers(1, n + 1))].copy()
    return heap


@pytest.mark.parametrize("seed", range(40))
def test_masks_equal_the_literal_walk(seed):
    rng = np.random.default_rng(7000 + seed)
    n_objects = int(rng.integers(1, 17))
    heap = random_heap(rng, n_objects, truncate=seed % 3 == 0)
    mesh_root = rng.integers(-2000, 2000, n_objects).astype(np.int32)
    mesh_root[rng.random(n_objects) < 0.15] = EMPTY_ROOT            # meshes without triangles are never tested
    words, _ = build_table(heap, mesh_root)
    assert len(words) >= 80
    n = len(heap)
    patterns = (np.array([[(k >> b) & 1 for b in range(n)] for k in range(1 << n)], bool) if n <= 11 else rng.random((4096, n)) < rng.uniform(0.3, 0.9))
    levels = int(words[1])
    N = (1 << levels) - 1
    for hit in patterns:
        full = np.zeros(N, bool); full[:n] = hit
        f_ref, t_ref = literal_walk(heap, hit, n_objects, mesh_root)
        f_got, t_pos, pos_tab, pos_of = masked_walk(words, full, heap)
        assert f_got == f_ref, (seed, hit, f_got, f_ref)
        # positions -> objects: pos_tab carries the roots; compare roots (ids can repeat in a random heap, roots identify the tests)
        inv = {p: i for i, p in pos_of.items()}
        got_ids = [int(heap[inv[p]]["index"]) for p in t_pos]
        assert got_ids == t_ref, (seed, hit, got_ids, t_ref)
        assert [int(pos_tab[p, 0]) for p in t_pos] == [int(mesh_root[i]) for i in t_ref]


def test_heaps_that_do_not_qualify_and_the_builders_heap():
    words, _ = build_table(np.zeros(0, scenes.BVHNODE_DT), np.zeros(3, np.int32))
    assert len(words) == 0
    big = np.zeros(63, scenes.BVHNODE_DT)
    words, _ = build_table(big, np.zeros(3, np.int32))
    assert len(words) == 0                                           # > 31 nodes: the kernel keeps the stack walk
    # the builder's own heap for 9 objects (C4's shape): 31 nodes, every leaf reachable, and only nodes that matter are evaluated
    rng = np.random.default_rng(3)
    lo = rng.uniform(-5, 5, (9, 3)).astype(np.float32)
    heap = scenes.build_object_bvh(lo, lo + 1)
    words, _ = build_table(heap, np.arange(9, dtype=np.int32))
    assert len(heap) == 31 and int(words[1]) == 5 and bin(int(words[5]) & 0xFFFFFFFF).count("1") == 9
    assert int(words[0]) == int((heap["vmin"] != heap["vmax"]).any(axis=1).sum())      # fillers are not evaluated
