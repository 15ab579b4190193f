"""CPU oracle for the -w path: the MinHash sketch (pure Python; small inputs only) and, r05, neighbour joining + the newick
text + the Tree class that stands in for ete3 when oracle/gen_golden.py runs the reference's GSC code (second half of the file).
TEST INFRASTRUCTURE ONLY.  Restates Mash 2.2's sketching (third-party, bundled as bin/mash; source
not vendored): canonical k-mer = min(k-mer, reverse complement) as strings, MurmurHash3_x64_128
(seed 42) of the ASCII k-mer, first 8 digest bytes, bottom-s distinct hashes.  Pinned by
tests/golden/mash.json (hash lists from `mash info -d`, distances from `mash dist`)."""

M64 = (1 << 64) - 1


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _fmix(k):
    k ^= k >> 33
    k = (k * 0xff51afd7ed558ccd) & M64
    k ^= k >> 33
    k = (k * 0xc4ceb9fe1a85ec53) & M64
    k ^= k >> 33
    return k


def murmur3_x64_128(data, seed):
    h1 = h2 = seed
    c1, c2 = 0x87c37b91114253d5, 0x4cf5ad432745937f
    n = len(data)
    nb = n // 16
    for i in range(nb):
        k1 = int.from_bytes(data[16 * i:16 * i + 8], "little")
        k2 = int.from_bytes(data[16 * i + 8:16 * i + 16], "little")
        k1 = (k1 * c1) & M64; k1 = _rotl(k1, 31); k1 = (k1 * c2) & M64; h1 ^= k1
        h1 = _rotl(h1, 27); h1 = (h1 + h2) & M64; h1 = (h1 * 5 + 0x52dce729) & M64
        k2 = (k2 * c2) & M64; k2 = _rotl(k2, 33); k2 = (k2 * c1) & M64; h2 ^= k2
        h2 = _rotl(h2, 31); h2 = (h2 + h1) & M64; h2 = (h2 * 5 + 0x38495ab5) & M64
    tail = data[16 * nb:]
    if len(tail) > 8:
        k2 = int.from_bytes(tail[8:], "little")
        k2 = (k2 * c2) & M64; k2 = _rotl(k2, 33); k2 = (k2 * c1) & M64; h2 ^= k2
    if len(tail) > 0:
        k1 = int.from_bytes(tail[:8], "little")
        k1 = (k1 * c1) & M64; k1 = _rotl(k1, 31); k1 = (k1 * c2) & M64; h1 ^= k1
    h1 ^= n; h2 ^= n
    h1 = (h1 + h2) & M64; h2 = (h2 + h1) & M64
    h1 = _fmix(h1); h2 = _fmix(h2)
    h1 = (h1 + h2) & M64; h2 = (h2 + h1) & M64
    return h1, h2


_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def sketch(fasta_bytes, k=21, sketch_size=1000, seed=42):
    """Bottom-s sketch of a plain FASTA (records separated at '>'; any non-ACGT letter breaks)."""
    hashes = set()
    for rec in fasta_bytes.decode().split(">")[1:]:
        seq = "".join(rec.split("\n")[1:]).upper()
        for i in range(len(seq) - k + 1):
            f = seq[i:i + k]
            if any(ch not in _COMP for ch in f):
                continue
            r = "".join(_COMP[ch] for ch in reversed(f))
            h = murmur3_x64_128(min(f, r).encode(), seed)[0]
            hashes.add(h if k > 16 else h & 0xffffffff)
    return sorted(hashes)[:sketch_size]


# --------------------------------------------------------------------------------------------
# Neighbour joining + the newick text between the tree and the GSC recursion (modeling.py:447-458, :465).
# Biopython 1.76 / ete3 3.1.1 are absent from the build container: restated from their published behaviour,
# PARITY UNPINNED (tie-breaking of the joins, "%1.5f" branch lengths, how ete3 reads a missing length).
# The arithmetic AFTER the parse -- clip_branch_lengths / set_branch_sum / set_node_weight (:478-503) -- is the
# reference's own code and is pinned: tests/golden/gsc_kat.json holds what those functions returned on trees parsed
# by the Tree class below (oracle/gen_golden.py::gen_gsc_kat hands it to the shim in ete3.Tree's place).
# --------------------------------------------------------------------------------------------
def nj_merges(mat):
    """orc_nj (psk_oracle.c section 7): plain O(n^3) neighbour joining.  mat: n x n symmetric, n >= 3.
    Returns (mi int32[n-2], mj int32[n-2], d1 f64[n-2], d2 f64[n-2], last) -- the joins as indices into the current
    clade list, the two branch lengths of each, and the distance between the two clades left at the end."""
    import ctypes
    import os
    import subprocess
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "libpsk_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", here, "libpsk_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(so)
    L.orc_nj.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_double)]
    L.orc_nj.restype = ctypes.c_int
    m = np.ascontiguousarray(np.asarray(mat, dtype=np.float64))
    n = m.shape[0]
    assert m.shape == (n, n) and n >= 3
    mi, mj = np.zeros(n - 2, np.int32), np.zeros(n - 2, np.int32)
    d1, d2 = np.zeros(n - 2), np.zeros(n - 2)
    last = ctypes.c_double()
    rc = L.orc_nj(n, m.ctypes.data, mi.ctypes.data, mj.ctypes.data, d1.ctypes.data, d2.ctypes.data, ctypes.byref(last))
    assert rc == 0, rc
    return mi, mj, d1, d2, last.value


class Tree:
    """The few members of ete3.Tree that the reference's GSC code touches (modeling.py:465-503): dist, name, up,
    get_children(), traverse("levelorder"), iter_leaves(); built from newick text of `format=1` (internal node names,
    branch lengths) or from the path of a file holding it.  OURS, not ete3's: a node without a length gets ete3's
    documented defaults (1.0; the root 0.0)."""

    def __init__(self, newick=None, format=1, name="", dist=1.0):
        self.name, self.dist, self.up, self.children = name, dist, None, []
        if newick is not None:
            import os
            text = open(newick).read() if os.path.exists(newick) else newick
            self._parse(text.strip())

    def _parse(self, s):
        assert s.endswith(";"), "newick must end with ';'"
        self.dist = 0.0
        stack = []
        cur = self
        i, n = 0, len(s) - 1
        while i < n:
            ch = s[i]
            if ch == "(":
                child = Tree()
                child.up = cur
                cur.children.append(child)
                stack.append(cur)
                cur = child
                i += 1
            elif ch == ",":
                parent = stack[-1]
                child = Tree()
                child.up = parent
                parent.children.append(child)
                cur = child
                i += 1
            elif ch == ")":
                cur = stack.pop()
                i += 1
            else:
                j = i
                while j < n and s[j] not in "(),":
                    j += 1
                label = s[i:j]
                if ":" in label:
                    nm, ln = label.rsplit(":", 1)
                    cur.dist = float(ln)
                else:
                    nm = label
                cur.name = nm
                i = j
        assert not stack, "unbalanced newick"

    def get_children(self):
        return list(self.children)

    def traverse(self, strategy="levelorder"):
        assert strategy == "levelorder"
        queue = [self]
        while queue:
            nd = queue.pop(0)
            yield nd
            queue.extend(nd.children)

    def iter_leaves(self):
        stack = [self]
        while stack:
            nd = stack.pop()
            if not nd.children:
                yield nd
            else:
                stack.extend(reversed(nd.children))


def nj_newick(names, mat):
    """Biopython's nj + newick writer as restated above: the joins of orc_nj replayed into clades named Inner<t>, the last
    two joined under the newer inner node (branch lengths 0 / the last distance), written with "%1.5f" lengths."""
    n = len(names)
    assert n >= 3
    mi, mj, d1, d2, last = nj_merges(mat)
    clades = [[nm, None, []] for nm in names]      # [name, branch length, children]
    inner = None
    for t in range(n - 2):
        a, b = clades[mi[t]], clades[mj[t]]
        a[1], b[1] = float(d1[t]), float(d2[t])
        inner = ["Inner%d" % (t + 1), None, [a, b]]
        clades[mj[t]] = inner
        del clades[mi[t]]
    if clades[0] is inner:
        clades[0][1], clades[1][1] = 0.0, last
        clades[0][2].append(clades[1])
        root = clades[0]
    else:
        clades[0][1], clades[1][1] = last, 0.0
        clades[1][2].append(clades[0])
        root = clades[1]

    def rec(c):
        kids = "(" + ",".join(rec(k) for k in c[2]) + ")" if c[2] else ""
        return "%s%s:%1.5f" % (kids, c[0], c[1])
    return rec(root) + ";"
