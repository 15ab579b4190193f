#!/usr/bin/env python3
"""A stand-in for PS_modeling_example_files.tar.gz (the reference's example set, /root/reference/example/test_PS_modeling.sh:12-17;
unreachable offline) with the same layout: a folder PS_modeling_example_files/ holding the FASTA files and data.pheno, whose
address column names the files relative to the directory the tarball is unpacked in (the example is run as
`phenotypeseeker modeling PS_modeling_example_files/data.pheno` from there).  Content: the 60 AT-rich (29 % GC, the GC share
of C. difficile), six-contig, 1-Mbp genomes of tests/golden/ds_atrich, regenerated from the parameters in its meta.json.
usage: tools/make_cfg1_standin.py OUT.tar.gz"""
import io
import json
import os
import sys
import tarfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402

out = sys.argv[1]
gd = os.path.join(ROOT, "tests", "golden", "ds_atrich")
with open(os.path.join(gd, "meta.json")) as f:
    meta = json.load(f)
gs = GenomeSet(**meta["synth"])
pheno = {l.split()[0]: l.split()[2] for l in open(os.path.join(gd, "data.pheno")).read().splitlines()[1:]}
rows = ["SampleID\tAddress\tAzithromycin"]
with tarfile.open(out, "w:gz", compresslevel=1) as t:
    for i in range(gs.n):
        name, fa = gs.sample(i)
        ti = tarfile.TarInfo("PS_modeling_example_files/%s.fasta" % name)
        ti.size = len(fa)
        t.addfile(ti, io.BytesIO(fa))
        rows.append("%s\tPS_modeling_example_files/%s.fasta\t%s" % (name, name, pheno[name]))
    blob = ("\n".join(rows) + "\n").encode()
    ti = tarfile.TarInfo("PS_modeling_example_files/data.pheno")
    ti.size = len(blob)
    t.addfile(ti, io.BytesIO(blob))
print(out, os.path.getsize(out), "bytes,", gs.n, "genomes")
