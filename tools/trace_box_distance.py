"""Where do the box distances on the root path of a point part?  (profiles/r05_notes.txt item 24.)

The case: seed 802 / case 760 of tools/fuzz_parity.py -- 60 000 points on a line in the plane, leaves of one point,
knn = 32 / 33, query row 26: the reference leaves out point 3505, which is the 32nd nearest by distance.

This script replays the reference search (kd_tree_search.hpp:52-105 + search_visitor.hpp:83-123) in numpy float32 on
the tree the oracle flattens, checks its row against the oracle's, and logs every far-child test.  Then it walks the
root path of the point in question and prints, per far child on it: the box distance that depends on the path alone
(what ANY search that carries {nbd, off[]} down the path computes -- the cooperative search included), the reference's
own value at its test, and the reference's bound `max()` at that moment.

    python tools/trace_box_distance.py [k] [row] [point]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle  # noqa: E402

f32 = np.float32


def the_case():
    rng = np.random.default_rng([802, 760])
    dim = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7]))
    n = int(rng.choice([1, 2, 9, 100, 3000, 20000, 60000]))
    nq = int(rng.choice([1, 63, 64, 65, 1000, 5000]))
    leaf = int(rng.choice([1, 2, 5, 10, 16, 24]))
    kind = str(rng.choice(["uniform", "clustered", "lattice", "duplicates", "line", "plane"]))
    scale = float(rng.choice([1.0, 1.0, 1e-6, 1e6, 37.5]))
    shift = float(rng.choice([0.0, 0.0, -0.5, 100.0]))
    rng.choice(["L2Squared", "L2Squared", "L1", "LPInf"])
    assert (dim, n, nq, leaf, kind, scale, shift) == (2, 60000, 64, 1, "line", 37.5, 0.0)
    pts = (((rng.random((n, 1)) * rng.random((1, dim)) + 0.25) + shift) * scale).astype(np.float32)
    assert rng.random() < 0.5
    q = (((rng.random((nq, 1)) * rng.random((1, dim)) + 0.25) + shift) * scale).astype(np.float32)
    return pts, q, leaf


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 33
    row = int(sys.argv[2]) if len(sys.argv) > 2 else 26
    target = int(sys.argv[3]) if len(sys.argv) > 3 else 3505
    pts, q, leaf = the_case()
    ora = oracle.Oracle(pts, leaf, "port")
    nodes, indices, _, _, depth = ora.flatten()
    want = ora.search_knn(q[row:row + 1], k)[0]
    qv = q[row]
    dim = pts.shape[1]
    fl = nodes.view(np.float32)
    print(f"tree: {len(nodes)} nodes, depth {depth}; query row {row} = {qv}; k = {k}")

    sys.setrecursionlimit(10000)
    lst = []  # sorted (distance, index), strict-< insertion: search_visitor.hpp:24-38
    off = np.zeros(dim, dtype=np.float32)
    tests = {}  # far child node -> (nbd', max() at the test, entered)
    entered_nbd = {0: f32(0)}
    parent = {}

    def vmax():
        return lst[k - 1][0] if len(lst) >= k else f32(np.finfo(np.float32).max)

    def visit(idx, d):
        if vmax() > d:
            j = len(lst)
            while j > 0 and d < lst[j - 1][0]:
                j -= 1
            lst.insert(j, (d, idx))
            del lst[k:]

    def search(n, nbd):
        if nodes[n, 2] == 0xFFFFFFFF:
            for i in range(int(nodes[n, 0]), int(nodes[n, 1])):
                p = pts[indices[i]]
                d = f32(0)
                for a in range(dim):
                    t = f32(qv[a] - p[a])
                    d = f32(d + f32(t * t))
                visit(int(indices[i]), d)
            return
        lm, rm = fl[n, 0], fl[n, 1]
        sd = int(nodes[n, 3])
        v = qv[sd]
        left, right = n + 1, int(nodes[n, 2])
        if f32(f32(f32(lm + rm) - v) - v) > 0:
            first, second = left, right
            t = f32(rm - v)
        else:
            first, second = right, left
            t = f32(lm - v)
        new_off = f32(t * t)
        parent[first] = (n, False, sd, new_off)
        parent[second] = (n, True, sd, new_off)
        entered_nbd[first] = nbd
        search(first, nbd)
        old = off[sd]
        nbd2 = f32(f32(nbd - old) + new_off)
        m = vmax()
        go = m >= nbd2
        tests[second] = (nbd2, m, go)
        if go:
            off[sd] = new_off
            entered_nbd[second] = nbd2
            search(second, nbd2)
            off[sd] = old

    search(0, f32(0))
    got = np.array([(i, d) for d, i in lst], dtype=want.dtype)
    assert got.tobytes() == want.tobytes(), "the numpy replay is not the oracle's row"
    print("numpy replay == oracle row;  in the row:", target in [i for _, i in lst])
    print("k-th distance:", repr(lst[-1][0]), " last entries:", lst[-4:])

    # the leaf of the target and its root path
    pos = int(np.nonzero(indices == target)[0][0])
    leaf_node = None
    n = 0
    path = [0]
    while nodes[n, 2] != 0xFFFFFFFF:
        # find which child holds position `pos`: the left subtree is n+1 .. right-1
        right = int(nodes[n, 2])
        # leftmost leaf begin of the right subtree
        r = right
        while nodes[r, 2] != 0xFFFFFFFF:
            r += 1
        n = n + 1 if pos < int(nodes[r, 0]) else right
        path.append(n)
    leaf_node = n
    p = pts[target]
    d = f32(0)
    for a in range(dim):
        t = f32(qv[a] - p[a])
        d = f32(d + f32(t * t))
    print(f"point {target}: leaf node {leaf_node}, depth {len(path) - 1}, distance {d!r}")

    # the path-determined state: what a search carrying {nbd, off[]} down this path computes
    poff = np.zeros(dim, dtype=np.float32)
    pnbd = f32(0)
    print(" depth  node      side  path nbd'        reference nbd'   reference max()   entered")
    for dpt in range(1, len(path)):
        a, c = path[dpt - 1], path[dpt]
        lm, rm = fl[a, 0], fl[a, 1]
        sd = int(nodes[a, 3])
        v = qv[sd]
        near_left = f32(f32(f32(lm + rm) - v) - v) > 0
        is_left = c == a + 1
        if is_left == near_left:
            continue  # the near child keeps its parent's state
        t = f32((lm if not is_left else rm) - v) if False else f32((rm if near_left else lm) - v)
        new_off = f32(t * t)
        pnbd = f32(f32(pnbd - poff[sd]) + new_off)
        poff[sd] = new_off
        rec = tests.get(c)
        if rec is None:
            print(f" {dpt:5d}  {c:8d}  far   {pnbd!r:16}  (never tested: an ancestor was not entered)")
            break
        flag = "" if rec[0] == pnbd else "   <-- DIFFERENT"
        print(f" {dpt:5d}  {c:8d}  far   {pnbd!r:16}  {rec[0]!r:16} {rec[1]!r:16}  {rec[2]}{flag}")
        if not rec[2]:
            print("        ^ the reference turned away here")
            break


if __name__ == "__main__":
    main()
