"""Reads a committed golden dataset directory (tests/golden/<tag>/) back for oracle/gen_golden.py."""
import gzip
import os


def load_dataset_files(d):
    """-> (names in data.pheno order, {name: inflated FASTA/FASTQ bytes})"""
    names, files = [], {}
    with open(os.path.join(d, "data.pheno")) as f:
        f.readline()
        for line in f:
            if line.strip():
                name, fn = line.split()[:2]
                names.append(name)
                with gzip.open(os.path.join(d, fn + ".gz"), "rb") as g:
                    files[name] = g.read()
    return names, files
