"""Command line of `phenotypeseeker` -- the reference's option surface (scripts/phenotypeseeker,
cli:31-337) kept flag for flag: same names, types, defaults and mutual exclusions, so existing
command lines run unchanged.  Options that select code outside the accelerated hot path are
accepted by the parser and rejected with a clear message by modeling.Input.Input_args."""
import argparse
import sys

# (flags, kwargs) per argument group of the `modeling` sub-command
_MODELING = {
    "Options for k-mer lists": [
        (("-l", "--kmer_length"), dict(type=str, metavar="", default="13",
                                        help="K-mer length. Must be an integer between (1-32, default = 13)")),
        (("-c", "--cutoff"), dict(type=str, metavar="INT", default="1",
                                   help="K-mer frequency cut-off. Must be an integer (default = 1).")),
    ],
    "Options for k-mer filtering by frequency": [
        (("--min",), dict(type=str, metavar="INT", default="0",
                           help="Minimum number of samples with support to report k-mer.")),
        (("--max",), dict(type=str, metavar="INT", default="0",
                           help="Maximum number of samples with support to report k-mer.")),
    ],
    "Options for k-mer filtering by pvalue": [
        (("--pvalue",), dict(metavar="", type=float, default=0.05,
                              help="P-value cut-off for k-mer filtering (default = 0.05)")),
        (("--n_kmers",), dict(metavar="", type=int, default=1000,
                               help="The maximum number of (lowest p-valued) k-mers selected for modelling "
                                    "(default = 1000). Set '0' for no limit")),
    ],
    "Options for regression models": [
        (("--pca",), dict(action="store_true", help="Apply PCA to k-mer features before fitting the model.")),
        (("--alphas",), dict(metavar="FLOAT", type=float, nargs="+",
                              help="List of alphas (regularization strengths) where to compute the models.")),
        (("--alpha_min",), dict(metavar="", type=float, default=1e-3, help="Start of the regularization path (1E-3).")),
        (("--alpha_max",), dict(metavar="", type=float, default=1e3, help="End of the regularization path (1E3).")),
        (("--n_alphas",), dict(metavar="", type=int, default=13, help="Number of alphas along the path (13).")),
        (("--gammas",), dict(metavar="FLOAT", type=float, nargs="+", help="List of gammas (rbf kernel).")),
        (("--gamma_min",), dict(metavar="", type=float, default=1e-3, help="Start of the gamma path.")),
        (("--gamma_max",), dict(metavar="", type=float, default=1e3, help="End of the gamma path.")),
        (("--n_gammas",), dict(metavar="", type=int, default=13, help="Number of gammas along the path.")),
        (("--n_iter",), dict(metavar="", type=int, default=25, help="Randomized-search iterations (rbf / RF).")),
        (("-cv2", "--n_splits_cv_inner"), dict(metavar="", type=int, default=10,
                                                 help="Number of folds for cross-validation of the training set. "
                                                      "Default min(10, no. samples in training set).")),
        (("--penalty",), dict(metavar="", type=str, default="l1", help="L1 (default), L2 or L1+L2")),
        (("-bc", "--binary_classifier"), dict(metavar="", type=str, default="log",
                                                choices=["log", "SVM", "RF", "NB", "XGBC", "DT"],
                                                help='The binary classifier: "log" (logistic regression; default) ...')),
        (("-reg", "--regressor"), dict(metavar="", type=str, default="lin", choices=["lin", "XGBR"],
                                        help='The regressor: "lin" (linear regression; default) ...')),
        (("--kernel",), dict(metavar="", type=str, default="linear", help="SVM kernel.")),
        (("--logreg_solver", "-ls"), dict(metavar="", type=str, default=None,
                                           help="Logistic regression solver ('liblinear' for L1).")),
        (("--l1_ratio",), dict(metavar="", type=float, default=0.5, help="The elastic net mixing parameter.")),
        (("-tow", "--train_on_whole"), dict(action="store_true",
                                             help="Train the output model on the whole dataset given.")),
        (("--max_iter",), dict(metavar="", type=float, default=1000,
                                help="Hard limit on iterations within solver (default = 1000).")),
        (("--tolerance", "-tol"), dict(metavar="", type=float, default=1e-4,
                                        help="Tolerance for stopping criterion (default = 1e-4).")),
    ],
    "Other options": [
        (("--kmerDB",), dict(metavar="", type=str, default=None,
                              help="Resistance database (FASTA) for k-mer filtering: k-mers absent from it are dropped.")),
        (("--omit_B_correction",), dict(action="store_true",
                                         help="Omit the Bonferroni multiple testing correction in k-mer filtering")),
        (("--mpheno",), dict(metavar="INT", type=int, nargs="+",
                              help="Ordinal numbers of columns of phenotypes to analyze (default = all)")),
        (("-w", "--weights"), dict(action="store_true", help="Use samples GSC weights in statistical testing of k-mers.")),
        (("-rc", "--real_counts"), dict(action="store_true",
                                         help="Use the real counts of k-mers instead of presence/absence.")),
        (("-a", "--assembly"), dict(action="store_true", help="Assemble the k-mers used in regression model.")),
        (("--take_logs",), dict(action="store_true", help="Take logarithms (base 2) of the phenotype values.")),
        (("-nt", "--num_threads"), dict(type=int, metavar="INT", default=8,
                                         help="Accepted for compatibility; the GPU engine is driven by one process.")),
        (("-jt", "--jump_to"), dict(type=str, metavar="", default=None, choices=["modelling", "modeling", "PCA"],
                                     help="Continue a discontinued process from the saved <pheno>_MLdf.csv.")),
    ],
}


def build_parser():
    parser = argparse.ArgumentParser(usage="PhenotypeSeeker {modeling,prediction} <INPUTFILE(S)> [OPTIONS]")
    parser.add_argument("--version", action="version", version="%(prog)s 1.2.3 (MI355X engine)")
    sub = parser.add_subparsers()
    pa = sub.add_parser("modeling", help="Generate phenotype prediction model",
                        usage="PhenotypeSeeker modeling INPUTFILE [OPTIONS]")
    pa.add_argument("inputfile", help="Text file of tab separated list of sample IDs, corresponding Fasta/Fastq "
                                      "file addresses and corresponding phenotype values (one or more column).")
    split = None
    for title, opts in _MODELING.items():
        grp = pa.add_argument_group(title)
        for flags, kw in opts:
            grp.add_argument(*flags, **kw)
        if title == "Options for regression models":
            split = grp.add_mutually_exclusive_group()
            split.add_argument("-cv1", "--n_splits_cv_outer", metavar="", type=int, default=None,
                               help="Number of folds to split dataset into training and test set.")
            split.add_argument("-ts", "--testset_size", metavar="", type=float, default=None,
                               help="The size of the test set in 1-fold train/test data splitting.")
    from . import modeling
    pa.set_defaults(func=modeling.modeling)

    pb = sub.add_parser("prediction", help="Use the PhenotypeSeeker model to predict phenotypes from genome data",
                        usage="PhenotypeSeeker prediction INPUTFILE1 INPUTFILE2 [OPTIONS]")
    pb.add_argument("inputfile1", help="Tab separated list of sample IDs and Fasta/Fastq file addresses.")
    pb.add_argument("inputfile2", help="Tab separated list of phenotypes to predict and model (.pkl) addresses.")
    pb.add_argument("-c", type=int, metavar="INT", default=1, help="K-mer frequency cut-off (default = 1).")
    pb.add_argument("-nt", "--num_threads", type=int, metavar="INT", default=8,
                    help="Host threads that read and frame the samples ahead of the GPU.")
    from . import prediction
    pb.set_defaults(func=prediction.prediction)
    return parser


def main(argv=None):
    # PSK_GPUS=N (N > 1) and no launcher in sight: this process, before it has loaded anything that could touch the GPU,
    # starts one rank per GPU itself (launch.py) -- the reference's parallel axis, `-nt`, needs no outside launcher
    # either (modeling.py:1649-1663).  Under a launcher that exports RANK / WORLD_SIZE the ranks come here directly.
    # Only `modeling` is sharded.  `prediction` counts a fixed dictionary per sample on one GPU and writes
    # predictions_<name>.txt / log.txt without looking at RANK: N copies of it would race on the same files and do the work
    # N times; --help / --version need no rank at all.  The sub-command is read off the raw arguments, before
    # anything that could load libpsk is imported, so the parent of a fan-out stays free of the GPU.
    import os
    rest = list(sys.argv[1:] if argv is None else argv)
    try:
        n_gpus = int(os.environ.get("PSK_GPUS", "0") or 0)
    except ValueError:
        sys.exit("PSK_GPUS=%s: expected the number of GPUs" % os.environ.get("PSK_GPUS"))
    sub = next((a for a in rest if not a.startswith("-")), None)
    wants_help = any(a in ("-h", "--help", "--version") for a in rest)
    if n_gpus > 1 and "WORLD_SIZE" not in os.environ and sub == "modeling" and not wants_help:
        from . import launch
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = {"PYTHONPATH": root + os.pathsep + os.environ.get("PYTHONPATH", "")}
        sys.exit(launch.spawn_ranks(["-m", "phenotypeseeker_amd.cli"] + rest, n_gpus,
                                    share_gpu=os.environ.get("PSK_SHARE_GPU") == "1", env_extra=env,
                                    deadline_s=launch.launch_timeout(default=0)))    # (a modeling run has no deadline unless the user sets one)
    if n_gpus > 1 and "WORLD_SIZE" not in os.environ and sub == "prediction" and not wants_help:
        sys.stderr.write("PSK_GPUS=%d: `prediction` is not sharded, it runs on one GPU\n" % n_gpus)
    parser = build_parser()
    args = parser.parse_args(argv)
    if not hasattr(args, "func"):
        parser.error("too few arguments")
    if argv is None or getattr(main, "_is_process", False):
        # the CLI IS the process: its phase table (modeling.Phases) starts where the process did
        from .modeling import process_start_epoch
        args._t0 = process_start_epoch()
    args.func(args)


if __name__ == "__main__":
    main._is_process = True
    main(sys.argv[1:])
