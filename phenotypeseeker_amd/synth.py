"""Deterministic synthetic genome sets (SURVEY.md section 8(d) "Synthetic inputs").

ancestor = i.i.d. uniform ACGT of length L; sample i = ancestor with per-site substitution
probability `sub_rate`; phenotype 1 for even i; a random "gene" of `gene_len` bases is
inserted at L/2 into phenotype-1 samples (1/12 dropout) and into 1/12 of the phenotype-0
samples.  Used by bench.py, the tests and oracle/gen_golden.py (there is no network and the
reference ships no data).  numpy only.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def wrap_fasta(name, codes, width=70):
    """codes: uint8 array of 0..3 -> FASTA bytes with `width`-column lines."""
    seq = _ACGT[codes]
    n = len(seq)
    nfull, rem = divmod(n, width)
    out = np.empty(nfull * (width + 1) + (rem + 1 if rem else 0), dtype=np.uint8)
    if nfull:
        body = out[: nfull * (width + 1)].reshape(nfull, width + 1)
        body[:, :width] = seq[: nfull * width].reshape(nfull, width)
        body[:, width] = 10
    if rem:
        out[nfull * (width + 1): -1] = seq[nfull * width:]
        out[-1] = 10
    return b">" + name.encode() + b"\n" + out.tobytes()


class GenomeSet:
    """Lazy generator: sample(i) -> (name, fasta_bytes); phenotype(i) -> 0/1."""

    def __init__(self, n_samples, length, seed=12345, sub_rate=0.003, gene_len=2000, dropout=12, gc=0.5, contigs=1):
        """gc: G+C fraction of the ancestor and of the substituted bases (0.5 = uniform ACGT; C. difficile, the
        reference's example organism, is at 0.29); contigs: FASTA records per sample (the genome is cut into that
        many pieces of unequal length, as an assembly would be)."""
        self.n = int(n_samples)
        self.length = int(length)
        self.seed = int(seed)
        self.sub_rate = float(sub_rate)
        self.dropout = int(dropout)
        self.gc = float(gc)
        self.contigs = int(contigs)
        rng = np.random.default_rng(self.seed)
        self.ancestor = self._bases(rng, self.length)
        self.gene = self._bases(rng, int(gene_len))

    def _bases(self, rng, n):
        if self.gc == 0.5:
            return rng.integers(0, 4, n, dtype=np.uint8)       # the draw the committed fixtures were made with
        at, gc = (1.0 - self.gc) / 2.0, self.gc / 2.0
        return rng.choice(4, size=n, p=[at, gc, gc, at]).astype(np.uint8)   # A C G T

    def name(self, i):
        return "S%04d" % i

    def phenotype(self, i):
        return 1 if i % 2 == 0 else 0

    def has_gene(self, i):
        ph = self.phenotype(i)
        j = i // 2
        if ph == 1:
            return (j % self.dropout) != self.dropout - 1
        return (j % self.dropout) == 0

    def codes(self, i):
        rng = np.random.default_rng([self.seed, 1 + i])
        g = self.ancestor.copy()
        nsub = rng.binomial(self.length, self.sub_rate)
        pos = rng.integers(0, self.length, nsub)
        g[pos] = self._bases(rng, nsub)
        if self.has_gene(i):
            h = self.length // 2
            g = np.concatenate([g[:h], self.gene, g[h:]])
        return g

    def sample(self, i):
        codes = self.codes(i)
        if self.contigs <= 1:
            return self.name(i), wrap_fasta(self.name(i) + "_c1", codes)
        rng = np.random.default_rng([self.seed, 7919, i])
        cuts = np.sort(rng.choice(np.arange(1, len(codes)), size=self.contigs - 1, replace=False))
        parts = np.split(codes, cuts)
        # assembler-style headers: name, length and coverage fields after a space
        return self.name(i), b"".join(wrap_fasta("%s_c%d len=%d cov=%.1f" % (self.name(i), j + 1, len(p), 30 + j), p)
                                      for j, p in enumerate(parts))

    def continuous_phenotype(self, i):
        """cfg 4: 2^(gene present) x lognormal noise."""
        rng = np.random.default_rng([self.seed, 100003, i])
        return float((2.0 if self.has_gene(i) else 1.0) * np.exp(rng.normal(0.0, 0.25)))


def fastq_reads(codes, n_reads, read_len, seed, err=0.005):
    """cfg 5: reads sampled uniformly from a genome, substitution errors, constant Phred 'I'."""
    rng = np.random.default_rng(seed)
    L = len(codes)
    starts = rng.integers(0, max(L - read_len, 1), n_reads)
    idx = starts[:, None] + np.arange(read_len)[None, :]
    reads = codes[np.minimum(idx, L - 1)].copy()
    mask = rng.random(reads.shape) < err
    reads[mask] = rng.integers(0, 4, int(mask.sum()), dtype=np.uint8)
    seqs = _ACGT[reads]
    rec_len = 0
    chunks = []
    qual = b"I" * read_len
    for r in range(n_reads):
        chunks.append(b"@r%d\n" % r + seqs[r].tobytes() + b"\n+\n" + qual + b"\n")
        rec_len += 1
    return b"".join(chunks)
