"""`-w/--weights`: Gerstein-Sonnhammer-Chothia sample weights from Mash distances
(Samples.get_weights and helpers, modeling.py:392-503).

    sketches            psk_minhash_sketch on the GPU            (was `mash sketch -r`, :386-390)
    mash_distance       Jaccard of two bottom-s sketches -> Mash distance, rounded to the 6
                        significant digits `mash dist` prints and the reference re-parses (:411-421)
    distance_matrix     the N x N table with the reference's row labelling (:415-428)
    nj                  neighbour joining as Bio.Phylo.TreeConstruction.DistanceTreeConstructor.nj
                        builds it, branch lengths then rounded like the newick writer ("%1.5f") the
                        reference round-trips through (:447-458)
    gsc_weights         clip to [1e-9, 1e9], BranchSum / NodeWeight recursion, mean-1 scaling (:461-503)

Parity: sketches and distances are pinned against the bundled mash binary
(tests/golden/mash.json).  Biopython 1.76 / ete3 3.1.1 are not installed in the build container,
so the tree step is restated from their documented behaviour: PARITY UNPINNED for nj + newick
rounding.
"""
import math


def mash_distance(a, b, k, sketch_size):
    """a, b: ascending distinct hash lists (bottom-s sketches).  Returns (distance, shared, denom)."""
    i = j = common = denom = 0
    na, nb = len(a), len(b)
    while denom < sketch_size and i < na and j < nb:
        if a[i] < b[j]:
            i += 1
        elif a[i] > b[j]:
            j += 1
        else:
            i += 1
            j += 1
            common += 1
        denom += 1
    if denom < sketch_size:
        if i < na:
            denom += na - i
        if j < nb:
            denom += nb - j
        if denom > sketch_size:
            denom = sketch_size
    if denom == 0:
        return 1.0, 0, 0
    jac = common / denom
    if common == denom:
        d = 0.0
    elif common == 0:
        d = 1.0
    else:
        d = -math.log(2.0 * jac / (1.0 + jac)) / k
    return float("%g" % d), common, denom  # `mash dist` prints 6 significant digits


def distances_from_counts(common, denom, k):
    """(shared, denominator) of every pair -> Mash distances as `mash dist` prints them (6 significant digits)."""
    import numpy as np
    common = np.asarray(common, dtype=np.float64)
    denom = np.asarray(denom, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        jac = common / denom
        d = -np.log(2.0 * jac / (1.0 + jac)) / k
    d = np.where(common == denom, 0.0, np.where(common == 0, 1.0, d))
    d = np.where(denom == 0, 1.0, d)
    # the 6-digit rounding goes through text as mash's printf does; a distance depends on (shared, denominator)
    # only, so each distinct value is formatted once
    uniq, inverse = np.unique(d.ravel(), return_inverse=True)
    return np.array([float("%g" % v) for v in uniq], dtype=np.float64)[inverse].reshape(d.shape)


def distance_matrix(names, sketches, k=21, sketch_size=1000, ctx=None):
    """names: sample names in data.pheno order; sketches: dict name -> hashes.
    `mash paste reference.msh K-mer_lists/*.msh` orders the sketches by FILE NAME (shell glob), while
    the reference labels the rows of the resulting table in data.pheno order (:415-428): when the
    pheno file is not sorted by sample name the labels are permuted.  Reproduced here.
    With an engine context the N(N+1)/2 sketch merges run on the GPU (psk_mash_pairs) -- what modeling.py
    does; the host loop below serves the CPU-only tests."""
    by_file = glob_order(names)
    n = len(names)
    if ctx is not None:
        common, denom = ctx.mash_pairs([sketches[nm] for nm in by_file], sketch_size)
        return list(names), distances_from_counts(common, denom, k)  # numpy array: nj() takes it as it is
    mat = [[0.0] * n for _ in range(n)]
    for r in range(n):
        for c in range(r + 1):
            d = mash_distance(sketches[by_file[r]], sketches[by_file[c]], k, sketch_size)[0]
            mat[r][c] = mat[c][r] = d
    return list(names), mat


def glob_order(names):
    """The order in which `mash paste reference.msh K-mer_lists/*.msh` (:406) takes the sketches: the shell sorts the
    FILE names, <name>.msh, bytewise (C locale) -- "S1-2.msh" comes before "S1.msh" although "S1" < "S1-2"."""
    return sorted(names, key=lambda nm: (nm + ".msh").encode())


def lower_triangle(mat):
    """The rows Bio's _DistanceMatrix takes: row i = its first i + 1 entries (_distance_matrix_modifier, :431-444)."""
    return [[float(mat[i][j]) for j in range(i + 1)] for i in range(len(mat))]


def distances_mat_text(labels, mat):
    """`distances.mat` as _mash_output_to_distance_matrix writes it (:415-428): per row the label, then a tab and the
    distance as `mash dist` printed it (C++ stream default: "%g") for every column; rows joined by a newline, none at
    the end."""
    import numpy as np
    m = np.asarray(mat, dtype=np.float64).reshape(len(labels), len(labels))
    uniq, inverse = np.unique(m.ravel(), return_inverse=True)        # (a million cells hold a few thousand distinct values)
    cells = np.array(["\t%g" % v for v in uniq], dtype=object)[inverse].reshape(m.shape)
    return "\n".join(labels[i] + "".join(cells[i]) for i in range(len(labels)))


class _Node:
    __slots__ = ("name", "dist", "children", "up", "BranchSum", "NodeWeight")

    def __init__(self, name, dist=0.0):
        self.name, self.dist, self.children, self.up = name, dist, [], None

    def add(self, child):
        child.up = self
        self.children.append(child)


def nj(names, mat, ctx=None):
    """Neighbour joining with Biopython's tie-breaking and rooting conventions (see module doc).  With an engine
    context the joins are computed on the GPU (psk_nj_merges) and only replayed here; the numpy loop below is
    the same arithmetic on the host (CPU-only tests, and the check of the kernel)."""
    n = len(names)
    clades = [_Node(nm) for nm in names]
    if ctx is not None and 3 <= n <= 4096:
        mi_a, mj_a, d1_a, d2_a, last = ctx.nj_merges(mat)
        inner = None
        for t in range(n - 2):
            mi, mj = int(mi_a[t]), int(mj_a[t])
            c1, c2 = clades[mi], clades[mj]
            inner = _Node("Inner%d" % (t + 1))
            inner.add(c1)
            inner.add(c2)
            c1.dist, c2.dist = float(d1_a[t]), float(d2_a[t])
            clades[mj] = inner
            del clades[mi]
        return _nj_root(clades, inner, last)
    dm = [[float(mat[i][j]) for j in range(n)] for i in range(n)]
    if n == 1:
        return clades[0]
    if n == 2:
        root = _Node("Inner")
        clades[1].dist = dm[1][0] / 2.0
        clades[0].dist = dm[1][0] - clades[1].dist
        root.add(clades[1])
        root.add(clades[0])
        return root
    import numpy as np
    dm = np.array(dm, dtype=np.float64)
    lower = np.tril(np.ones((n, n), dtype=bool), -1)
    inner = None
    count = 0
    m = n
    while m > 2:
        d = dm[:m, :m]
        # the same IEEE operations in the same order as the scalar loops they replace: left-to-right row
        # sums, (d[i][j] - nd[i]) - nd[j], first minimum in (i ascending, j < i ascending) scan order
        node_dist = np.cumsum(d, axis=1)[:, -1] / (m - 2)
        t = (d - node_dist[:, None]) - node_dist[None, :]
        flat = int(np.argmin(np.where(lower[:m, :m], t, np.inf)))
        mi, mj = divmod(flat, m)
        if (mi, mj) == (1, 0):  # the scan starts from this pair with the indices the other way round
            mi, mj = 0, 1
        c1, c2 = clades[mi], clades[mj]
        count += 1
        inner = _Node("Inner%d" % count)
        inner.add(c1)
        inner.add(c2)
        c1.dist = float((d[mi, mj] + node_dist[mi] - node_dist[mj]) / 2.0)
        c2.dist = float(d[mi, mj] - c1.dist)
        clades[mj] = inner
        del clades[mi]
        v = (d[mi, :] + d[mj, :] - d[mi, mj]) / 2.0
        keep = np.ones(m, dtype=bool)
        keep[mi] = keep[mj] = False
        d[mj, keep] = v[keep]
        d[keep, mj] = v[keep]
        # delete row and column mi in place (order of the rest preserved)
        dm[mi:m - 1, :m] = dm[mi + 1:m, :m].copy()
        dm[:m - 1, mi:m - 1] = dm[:m - 1, mi + 1:m].copy()
        m -= 1
    return _nj_root(clades, inner, float(dm[1, 0]))


def _nj_root(clades, inner, last):
    """The library's last step: the two remaining clades are joined, the newer inner node becomes the root."""
    if clades[0] is inner:
        clades[0].dist = 0.0
        clades[1].dist = last
        clades[0].add(clades[1])
        return clades[0]
    clades[0].dist = last
    clades[1].dist = 0.0
    clades[1].add(clades[0])
    return clades[1]


def _walk(node, order="pre"):
    """Depth first, children left to right; without recursion (a neighbour-joining tree of a clonal set is a
    caterpillar as deep as it has leaves)."""
    stack = [(node, 0)]
    while stack:
        nd, k = stack.pop()
        if k == 0 and order == "pre":
            yield nd
        if k < len(nd.children):
            stack.append((nd, k + 1))
            stack.append((nd.children[k], 0))
        elif order == "post":
            yield nd


def newick_round_trip(root):
    """Branch lengths as they come back from the phyloxml -> newick ("%1.5f") -> ete3 round trip."""
    for nd in _walk(root):
        nd.dist = float("%1.5f" % (nd.dist if nd.dist is not None else 0.0))
    return root


def to_newick(root):
    text = {}
    for nd in _walk(root, "post"):
        inner = "(" + ",".join(text.pop(id(c)) for c in nd.children) + ")" if nd.children else ""
        text[id(nd)] = "%s%s:%1.5f" % (inner, nd.name, nd.dist)
    return text[id(root)] + ";"


def from_newick(text):
    """Newick text with internal node names and branch lengths (what Bio.Phylo.convert leaves in tree_newick.txt and
    ete3.Tree(path, format=1) reads back, :455-465) -> the tree of _Node.  A node without a length gets ete3's
    defaults: 1.0, the root 0.0."""
    text = text.strip()
    if not text.endswith(";"):
        raise ValueError("newick text must end with ';'")
    root = cur = _Node("", 0.0)
    stack = []
    i, n = 0, len(text) - 1
    while i < n:
        ch = text[i]
        if ch == "(" or ch == ",":
            if ch == "(":
                stack.append(cur)
            elif not stack:
                raise ValueError("newick: ',' outside of parentheses")
            child = _Node("", 1.0)
            stack[-1].add(child)
            cur = child
            i += 1
        elif ch == ")":
            if not stack:
                raise ValueError("newick: unbalanced ')'")
            cur = stack.pop()
            i += 1
        else:
            j = i
            while j < n and text[j] not in "(),":
                j += 1
            label = text[i:j]
            if ":" in label:
                label, length = label.rsplit(":", 1)
                cur.dist = float(length)
            cur.name = label.strip()
            i = j
    if stack:
        raise ValueError("newick: unbalanced '('")
    return root


def gsc_weights(root, min_val=1e-9, max_val=1e9):
    """Leaf name -> GSC weight scaled to mean 1 (GSC_weights_from_newick(normalize='mean1'), :461-503).  The same
    IEEE operations in the same order as clip_branch_lengths / set_branch_sum / set_node_weight: a node's BranchSum is
    ((0 + c1.BranchSum) + c1.dist) + c2.BranchSum ... over its children left to right; pinned bit for bit by
    tests/golden/gsc_kat.json (the reference's functions on the same trees)."""
    for nd in _walk(root):
        if nd.dist > max_val:
            nd.dist = max_val
        elif nd.dist < min_val:
            nd.dist = min_val
    for nd in _walk(root, "post"):
        total = 0
        for c in nd.children:
            total += c.BranchSum
            total += c.dist
        nd.BranchSum = total
    for nd in _walk(root):
        if nd.up is None:
            nd.NodeWeight = 1.0
        else:
            nd.NodeWeight = nd.up.NodeWeight * (nd.dist + nd.BranchSum) / nd.up.BranchSum
    weights = {}
    for nd in _walk(root):
        if not nd.children:
            weights[nd.name] = nd.NodeWeight
    return {k: v * len(weights) for k, v in weights.items()}


def weights_from_sketches(names, sketches, k=21, sketch_size=1000, ctx=None, files_dir=None):
    """Samples.get_weights (:392-400).  files_dir: where to leave `distances.mat` and `tree_newick.txt`, the two
    intermediate files of the reference's chain that are its own formats (mash_distances.mat / reference.msh /
    tree_xml.txt are Mash's and Biopython's)."""
    import os
    labels, mat = distance_matrix(names, sketches, k, sketch_size, ctx)
    if files_dir is not None:
        with open(os.path.join(files_dir, "distances.mat"), "w") as f:
            f.write(distances_mat_text(labels, mat))
    tree = newick_round_trip(nj(labels, mat, ctx))
    if files_dir is not None:
        with open(os.path.join(files_dir, "tree_newick.txt"), "w") as f:
            f.write(to_newick(tree) + "\n")
    return gsc_weights(tree), tree
