"""`phenotypeseeker prediction` on the MI355X engine (mirror of /root/reference/PhenotypeSeeker/
prediction.py; line numbers cite it).  The `gmer_counter -db` subprocess per sample (:72-80) is
replaced by psk_count_dict; the k-mer text database and the per-sample count files are never
written.  The model file is the reference's joblib dict {'model', 'kmers', 'pca', 'pred_scale'}."""
import os
import sys
import time
from collections import OrderedDict

import numpy as np

from . import formats
from ._lib import PSK_EGZIP, PskError
from .engine import PskContext


def timer(f):
    def wrapper(*args):
        start = time.time()
        f(*args)
        with open("log.txt", "a") as log:
            log.write("Func %s took %s secs\n" % (f, time.time() - start))
    return wrapper


class Samples:
    no_samples = 0

    def __init__(self, name, address):
        self.name, self.address = name, address
        Samples.no_samples += 1

    @classmethod
    def from_inputfile(cls, line):
        fields = line.split()
        return cls(fields[0], fields[1])

    @classmethod
    def map_samples(cls, ctx, samples, pheno, n_threads):
        """was: Pool(num_threads).map over samples of `gmer_counter -db K-mer_lists/k-mer_db_<pheno>.txt <address>`
        (:72-80, :150-163) followed by the >= cutoff thresholding of kmer_counts (:82-100).  All samples go through
        one batched call: the library's threads read and frame the files ahead of the GPU (compressed inputs are
        inflated here first), one kernel per sample, one read-back."""
        paths = [s.address for s in samples]
        counts = None
        # .gz files go to the library as they are (r05: it inflates them, on the GPU when there is enough of them); with
        # PSK_NO_GPU_GZ=1 they are recognised here by their magic bytes (the suffix alone is not trusted) and inflated here
        zipped = bool(os.environ.get("PSK_NO_GPU_GZ")) and any(p.endswith(".gz") or formats.is_gzip(p) for p in paths)
        if not zipped:
            try:
                counts = ctx.count_dict_files(paths, pheno.k, pheno.words, n_threads)
            except PskError as exc:       # the library checks the magic bytes too (a file that changed under the probe)
                if exc.code != PSK_EGZIP:
                    raise
        if counts is None:
            counts = np.zeros((len(paths), len(pheno.words)), dtype=np.uint32)
            for lo in range(0, len(paths), 32):     # bounded memory: 32 inflated files at a time
                part = [formats.read_sequence_file(p) for p in paths[lo:lo + 32]]
                counts[lo:lo + 32] = ctx.count_dict_batch(part, pheno.k, pheno.words, n_threads)
        return (counts >= Phenotypes.cutoff).astype(np.float64)


class Phenotypes:
    cutoff = 1
    no_phenotypes = 0

    def __init__(self, name, model, kmers, pca, pred_scale):
        self.name, self.model, self.kmers, self.pca, self.pred_scale = name, model, kmers, pca, pred_scale
        self.k = len(str(kmers[0])) if kmers.shape[0] else 0
        self.words = np.array([formats.canonical(formats.kmer_to_word(str(km)), self.k) for km in kmers],
                              dtype=np.uint64)
        self.matrix = np.empty(shape=(Samples.no_samples, kmers.shape[0]))
        Phenotypes.no_phenotypes += 1

    @classmethod
    def from_inputfile(cls, line):
        """'<phenotype> <model.pkl>' (:123-143)"""
        from . import skpickle
        name, path = line.split()[0], line.split()[1]
        # a plain pickle of a linear model (what `modeling` writes) is read without importing joblib / scikit-learn
        # (0.5-2 s of a run that otherwise takes half a second); anything else goes through joblib.load as in the reference
        pkg = None if os.environ.get("PSK_JOBLIB_LOAD") else skpickle.load_linear_package(path)
        if pkg is None:
            import joblib
            pkg = joblib.load(path)
        if pkg.get("pca"):
            raise SystemExit("PCA models are outside the accelerated path.")
        return cls(name, pkg["model"], np.asarray(pkg["kmers"]), False, pkg["pred_scale"])

    def get_inp_matrix(self, ctx, n_threads=8):
        self.matrix[:, :] = Samples.map_samples(ctx, list(Input.samples.values()), self, n_threads)

    def predict(self):
        """predictions_<pheno>.txt (:165-182)"""
        predictions = self.model.predict(self.matrix)
        with open("predictions_" + self.name + ".txt", "w+") as out:
            if self.pred_scale == "binary":
                proba = self.model.predict_proba(self.matrix)
                out.write("Sample_ID\tpredicted_phenotype\tprobability_for_predicted_class\n")
                for sample, pred, pr in zip(Input.samples.keys(), predictions, proba):
                    out.write("%s\t%s\t%s\n" % (sample, str(pred), str(round(pr[1], 2))))
            else:
                out.write("Sample_ID\tpredicted_phenotype\n")
                for sample, pred in zip(Input.samples.keys(), predictions):
                    out.write(sample + "\t" + str(pred) + "\n")


class Input:
    samples = OrderedDict()
    phenos = OrderedDict()

    @classmethod
    def reset(cls):
        cls.samples, cls.phenos = OrderedDict(), OrderedDict()
        Samples.no_samples = 0
        Phenotypes.no_phenotypes = 0

    @classmethod
    def get_samples(cls, inputfile):
        with open(inputfile) as fh:
            for line in fh:
                if line.strip():
                    cls.samples[line.split()[0]] = Samples.from_inputfile(line)

    @classmethod
    def get_phenos(cls, inputfile):
        with open(inputfile) as fh:
            for line in fh:
                if line.strip():
                    cls.phenos[line.split()[0]] = Phenotypes.from_inputfile(line)


@timer
def prediction(args):
    sys.stderr.write("\x1b[1;1;101m######                   PhenotypeSeeker                   ######\x1b[0m\n")
    sys.stderr.write("\x1b[1;1;101m######                     prediction                      ######\x1b[0m\n\n")
    Input.reset()
    Phenotypes.cutoff = args.c
    Input.get_samples(args.inputfile1)
    Input.get_phenos(args.inputfile2)
    with PskContext(0) as ctx:
        for pheno in Input.phenos.values():
            sys.stderr.write("\x1b[1;32mPredicting the phenotypes for %s.\x1b[0m\n" % pheno.name)
            pheno.get_inp_matrix(ctx, max(1, min(int(getattr(args, "num_threads", 8) or 8), 16)))
            pheno.predict()
    sys.stderr.write("\n\x1b[1;1;101m######          PhenotypeSeeker prediction finished          ######\x1b[0m\n")
