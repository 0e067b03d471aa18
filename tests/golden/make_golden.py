#!/usr/bin/env python3
"""Regenerate tests/golden/{cases.json,synthetic/*} from the REAL reference libfastx (oracle/_ref/fxref).

Run in the build container only (needs /root/reference to have built oracle/_ref/fxref):
    python tests/golden/make_golden.py
Inputs come from the deterministic generator (oracle/fxoracle.c); expected outputs are
what the reference code printed.  Generated cases store md5 + counts (inputs are regenerated from the seed); the variable-length
cases, whose inputs are themselves reference output, store input and output text.
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import fxoracle_py as fo  # noqa: E402

AD = "AGATCGGAAGAGC"


def md5(b):
    return hashlib.md5(b).hexdigest()


def run_chain(text, chain):
    ref = fo.ref_binary()
    assert ref, "build oracle/_ref first: make -C oracle ref"
    for cmd in chain:
        p = subprocess.run([ref] + cmd, input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, (cmd, p.stderr)
        text = p.stdout
    return text


def stats(text):
    lines = text.split(b"\n")
    seqs = lines[1::4]
    return len([s for s in seqs if s]), sum(len(s) for s in seqs)


# name, generator (seed, n, L, adapter), chain of reference command lines, engine parameters, keep full text?
P = dict
CASES = [
    ("cfg1", (1, 100000, 36, False), [["fastq_quality_trimmer", "-t", "20", "-l", "30"]],
     P(stages=2, qt_threshold=20, qt_min_len=30), False),
    ("cfg2_200k", (2, 200000, 150, False), [["fastq_quality_trimmer", "-t", "20", "-l", "30"], ["fastq_quality_filter", "-q", "20", "-p", "80"]],
     P(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80), False),
    ("cfg2_1m", (2, 1000000, 150, False), [["fastq_quality_trimmer", "-t", "20", "-l", "30"], ["fastq_quality_filter", "-q", "20", "-p", "80"]],
     P(stages=6, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80), False),
    ("cfg3_200k", (3, 200000, 100, True), [["fastx_clipper", "-a", AD, "-l", "15", "-n"]],
     P(stages=1, adapter=AD, clip_min_len=15, clip_flags=4), False),
    ("cfg4_200k", (2, 200000, 150, False), [["fastx_reverse_complement"], ["fastx_trimmer", "-f", "5", "-l", "145"]],
     P(stages=24, ft_first=5, ft_last=145), False),
    ("cfg5_200k", (5, 200000, 150, True), [["fastx_clipper", "-a", AD, "-l", "15", "-n"], ["fastq_quality_trimmer", "-t", "20", "-l", "30"],
                                           ["fastq_quality_filter", "-q", "20", "-p", "80"]],
     P(stages=7, adapter=AD, clip_min_len=15, clip_flags=4, qt_threshold=20, qt_min_len=30, qf_min_quality=20, qf_min_percent=80), False),
    # small full-text cases: parameter corners of every tool
    ("qtrim_neg_t", (11, 1500, 75, False), [["fastq_quality_trimmer", "-t", "-3"]], P(stages=2, qt_threshold=-3), True),
    ("qtrim_t35_l60", (12, 1500, 75, False), [["fastq_quality_trimmer", "-t", "35", "-l", "60"]], P(stages=2, qt_threshold=35, qt_min_len=60), True),
    ("qfilter_no_p", (13, 1500, 75, False), [["fastq_quality_filter", "-q", "30"]], P(stages=4, qf_min_quality=30), True),
    ("qfilter_q94_no_p", (13, 300, 75, False), [["fastq_quality_filter", "-q", "94"]], P(stages=4, qf_min_quality=94), True),
    ("qfilter_q30_p100", (14, 1500, 50, False), [["fastq_quality_filter", "-q", "10", "-p", "100"]], P(stages=4, qf_min_quality=10, qf_min_percent=100), True),
    ("qfilter_q25_p50", (14, 1500, 50, False), [["fastq_quality_filter", "-q", "25", "-p", "50"]], P(stages=4, qf_min_quality=25, qf_min_percent=50), True),
    ("clip_default", (15, 1500, 60, True), [["fastx_clipper", "-a", AD]], P(stages=1, adapter=AD), True),
    ("clip_c", (15, 1500, 60, True), [["fastx_clipper", "-a", AD, "-c", "-n"]], P(stages=1, adapter=AD, clip_flags=1 | 4), True),
    ("clip_C", (15, 1500, 60, True), [["fastx_clipper", "-a", AD, "-C", "-n"]], P(stages=1, adapter=AD, clip_flags=2 | 4), True),
    ("clip_k", (15, 1500, 60, True), [["fastx_clipper", "-a", AD, "-k"]], P(stages=1, adapter=AD, clip_flags=8), True),
    ("clip_d5_M8", (16, 1500, 60, True), [["fastx_clipper", "-a", AD, "-d", "5", "-M", "8", "-n", "-l", "10"]],
     P(stages=1, adapter=AD, clip_keep_delta=5 + len(AD), clip_min_adapter_len=8, clip_min_len=10, clip_flags=4), True),
    ("clip_long_adapter", (17, 1200, 80, True), [["fastx_clipper", "-a", "TGGAATTCTCGGGTGCCAAGGAACTCCAGTCAC", "-n"]],
     P(stages=1, adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAGTCAC", clip_flags=4), True),
    ("clip_dummy_adapter", (18, 1200, 40, False), [["fastx_clipper"]], P(stages=1), True),
    ("ftrim_end", (19, 1500, 75, False), [["fastx_trimmer", "-t", "10", "-m", "30"]], P(stages=32, ft_trim_end=10, ft_min_len=30), True),
    ("ftrim_f20", (19, 1500, 75, False), [["fastx_trimmer", "-f", "20"]], P(stages=16, ft_first=20), True),
    ("ftrim_l20", (19, 1500, 75, False), [["fastx_trimmer", "-l", "20"]], P(stages=16, ft_last=20), True),
    ("revcomp", (20, 1500, 75, False), [["fastx_reverse_complement"]], P(stages=8), True),
    ("masker_q20_dot", (23, 1500, 75, False), [["fastq_masker", "-q", "20", "-r", "."]], P(stages=64, mask_min_quality=20, mask_char="."), True),
    ("masker_default", (23, 1500, 75, False), [["fastq_masker"]], P(stages=64), True),
    ("artifacts", (24, 3000, 36, False), [["fastx_artifacts_filter"]], P(stages=128), True),
]

# variable-length inputs: produced by quality-trimming a synthetic set first (the trimmed file is the INPUT)
VARLEN = [
    ("var_qfilter", (21, 3000, 100, False), [["fastq_quality_filter", "-q", "20", "-p", "80"]], P(stages=4, qf_min_quality=20, qf_min_percent=80)),
    ("var_revcomp_ftrim", (21, 3000, 100, False), [["fastx_reverse_complement"], ["fastx_trimmer", "-f", "20", "-l", "80"]],
     P(stages=24, ft_first=20, ft_last=80)),
    ("var_ftrim_end", (21, 3000, 100, False), [["fastx_trimmer", "-t", "25", "-m", "20"]], P(stages=32, ft_trim_end=25, ft_min_len=20)),
    ("var_clip_history", (22, 3000, 80, True), [["fastx_clipper", "-a", AD, "-n", "-c"]], P(stages=1, adapter=AD, clip_flags=1 | 4)),
]

GALAXY = [  # SURVEY.md section 4 (XML <tests> blocks)
    dict(name="galaxy_quality_trimmer", input="fastq_quality_trimmer.fastq", expect="fastq_quality_trimmer.out",
         cmd=["fastq_quality_trimmer", "-Q", "64", "-t", "30", "-l", "16"], params=P(stages=2, qoffset=64, qt_threshold=30, qt_min_len=16)),
    dict(name="galaxy_quality_filter_a", input="fastq_qual_filter1.fastq", expect="fastq_qual_filter1a.out",
         cmd=["fastq_quality_filter", "-Q", "64", "-q", "33", "-p", "100"], params=P(stages=4, qoffset=64, qf_min_quality=33, qf_min_percent=100)),
    dict(name="galaxy_quality_filter_b", input="fastq_qual_filter1.fastq", expect="fastq_qual_filter1b.out",
         cmd=["fastq_quality_filter", "-Q", "64", "-q", "20", "-p", "80"], params=P(stages=4, qoffset=64, qf_min_quality=20, qf_min_percent=80)),
    dict(name="galaxy_clipper", input="fastx_clipper1.fastq", expect="fastx_clipper1a.out",
         cmd=["fastx_clipper", "-Q", "64", "-l", "15", "-a", "CAATTGGTTAATCCCCCTATATA", "-d", "0", "-n", "-c"],
         params=P(stages=1, qoffset=64, adapter="CAATTGGTTAATCCCCCTATATA", clip_min_len=15, clip_flags=1 | 4)),
    dict(name="galaxy_trimmer_fasta", input="fastx_trimmer1.fasta", expect="fastx_trimmer1.out",
         cmd=["fastx_trimmer", "-f", "5", "-l", "36"], params=P(stages=16, ft_first=5, ft_last=36)),
    dict(name="galaxy_trimmer_numeric", input="fastx_trimmer2.fastq", expect="fastx_trimmer2.out",
         cmd=["fastx_trimmer", "-f", "1", "-l", "27"], params=P(stages=16, ft_first=1, ft_last=27)),
    dict(name="galaxy_trimmer_from_end", input="fastx_trimmer_from_end1.fasta", expect="fastx_trimmer_from_end1.out",
         cmd=["fastx_trimmer", "-t", "2", "-m", "16"], params=P(stages=32, ft_trim_end=2, ft_min_len=16)),
    dict(name="galaxy_revcomp_fasta", input="fastx_rev_comp1.fasta", expect="fastx_reverse_complement1.out",
         cmd=["fastx_reverse_complement"], params=P(stages=8)),
    dict(name="galaxy_revcomp_numeric", input="fastx_rev_comp2.fastq", expect="fastx_reverse_complement2.out",
         cmd=["fastx_reverse_complement"], params=P(stages=8)),
    # neighbouring per-read tools on the same batch ABI
    dict(name="galaxy_masker", input="fastq_masker.fastq", expect="fastq_masker.out",
         cmd=["fastq_masker", "-Q", "64", "-q", "29", "-r", "x"], params=P(stages=64, qoffset=64, mask_min_quality=29, mask_char="x")),
    dict(name="galaxy_artifacts_fasta", input="fastx_artifacts1.fasta", expect="fastx_artifacts1.out",
         cmd=["fastx_artifacts_filter"], params=P(stages=128)),
    dict(name="galaxy_fastq_to_fasta_a", input="fastq_to_fasta1.fastq", expect="fastq_to_fasta1a.out",
         cmd=["fastq_to_fasta", "-Q", "64"], params=P(stages=256, qoffset=64), fasta_out=True),
    dict(name="galaxy_fastq_to_fasta_b", input="fastq_to_fasta1.fastq", expect="fastq_to_fasta1b.out",
         cmd=["fastq_to_fasta", "-Q", "64", "-n", "-r"], params=P(stages=256, qoffset=64, nf_keep_n=1), fasta_out=True, rename=True),
    dict(name="galaxy_artifacts_numeric", input="fastx_artifacts2.fastq", expect="fastx_artifacts2.out",
         cmd=["fastx_artifacts_filter"], params=P(stages=128)),
]


def main():
    out = dict(galaxy=GALAXY, synthetic=[], varlen=[])
    syn = os.path.join(HERE, "synthetic")
    os.makedirs(syn, exist_ok=True)
    for g in GALAXY:  # the reference driver must agree with the reference's own golden files
        inp = open(os.path.join(HERE, "galaxy", g["input"]), "rb").read()
        exp = open(os.path.join(HERE, "galaxy", g["expect"]), "rb").read()
        assert run_chain(inp, [g["cmd"]]) == exp, g["name"]
    for name, (seed, n, L, ad), chain, params, full in CASES:
        text = fo.synth_fastq(seed, 0, n, L, ad)
        o = run_chain(text, chain)
        kept, bases = stats(o)
        rec = dict(name=name, seed=seed, n=n, L=L, adapter=ad, chain=chain, params=params, input_md5=md5(text), output_md5=md5(o),
                   kept=kept, kept_bases=bases, full=full)
        out["synthetic"].append(rec)
        print(name, kept, bases, rec["output_md5"])
    for name, (seed, n, L, ad), chain, params in VARLEN:
        text = run_chain(fo.synth_fastq(seed, 0, n, L, ad), [["fastq_quality_trimmer", "-t", "22", "-l", "12"]])
        o = run_chain(text, chain)
        kept, bases = stats(o)
        open(os.path.join(syn, name + ".fq"), "wb").write(text)
        open(os.path.join(syn, name + ".out"), "wb").write(o)
        out["varlen"].append(dict(name=name, chain=chain, params=params, kept=kept, kept_bases=bases, input_md5=md5(text), output_md5=md5(o)))
        print(name, kept, bases)
    json.dump(out, open(os.path.join(HERE, "cases.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
