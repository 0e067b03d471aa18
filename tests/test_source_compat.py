"""CPU tier, build container only: the reference's OWN tool sources compile unchanged against the host layer's headers.

INTEGRATION.md section 2 claims that `fastx_toolkit_amd/host/fastx.h` / `fastx_args.h` are source compatible with the reference's
`src/libfastx/fastx.h:62-142` and `fastx_args.h:27-38`: same type and field names, same prototypes, same argument meaning.  This test
is the guard of that claim.  For each reference tool below it compiles the file WHERE IT LIES under /root/reference (nothing is copied)
with `-I host/` in place of the reference's libfastx headers, links `host/bin/libfastx_amd.a`, runs the reference's Galaxy test pairs
through the resulting binary and compares the bytes.  The per-record API is host code (no GPU is touched).

The tools' sources include an autoconf-generated `config.h` for one macro (PACKAGE_STRING, used in the usage text); an empty `config.h`
written to tmp_path plus `-DPACKAGE_STRING=...` stands in for it HERE ONLY: this is a boundary test of our headers, it pins nothing about
the oracle (oracle/Makefile builds libfastx without any stand-in).  Nothing of this travels to the GPU box: /root/reference does not
exist there and the test skips.
"""
import json
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/src"
HOST = os.path.join(ROOT, "fastx_toolkit_amd", "host")
GAL = os.path.join(ROOT, "tests", "golden", "galaxy")

TOOLS = {       # reference source (compiled unchanged) -> names of the Galaxy cases in tests/golden/cases.json it must reproduce
    "fastx_trimmer/fastx_trimmer.c": ["galaxy_trimmer_fasta", "galaxy_trimmer_numeric", "galaxy_trimmer_from_end"],
    "fastq_quality_trimmer/fastq_quality_trimmer.c": ["galaxy_quality_trimmer"],
    "fastq_quality_filter/fastq_quality_filter.c": ["galaxy_quality_filter_a", "galaxy_quality_filter_b"],
    "fastx_reverse_complement/fastx_reverse_complement.c": ["galaxy_revcomp_fasta", "galaxy_revcomp_numeric"],
    "fastq_masker/fastq_masker.c": ["galaxy_masker"],
    "fastx_artifacts_filter/fastx_artifacts_filter.c": ["galaxy_artifacts_fasta", "galaxy_artifacts_numeric"],
}

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources exist in the build container only")


@pytest.fixture(scope="module")
def host_lib():
    from fastx_toolkit_amd import build as b
    b.build_all()
    a = os.path.join(HOST, "bin", "libfastx_amd.a")
    assert os.path.exists(a)
    return a


@pytest.mark.parametrize("src", sorted(TOOLS))
def test_reference_tool_source_compiles_unchanged_and_reproduces_its_galaxy_pairs(src, host_lib, tmp_path):
    cases = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "cases.json")))["galaxy"]}
    (tmp_path / "config.h").write_text("")                    # see the module docstring
    exe = str(tmp_path / os.path.basename(src)[:-2])
    cmd = ["gcc", "-O1", "-std=gnu11", "-DPACKAGE_STRING=\"FASTX Toolkit 0.0.14\"", "-I", str(tmp_path), "-I", HOST, os.path.join(REF, src), host_lib,
           "-L", os.path.join(ROOT, "fastx_toolkit_amd"), "-lfxg", "-lpthread", "-lz", "-Wl,-rpath," + os.path.join(ROOT, "fastx_toolkit_amd"), "-o", exe]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, "the reference's %s no longer compiles against host/fastx.h + fastx_args.h:\n%s" % (src, p.stdout[-3000:])
    for name in TOOLS[src]:
        c = cases[name]
        out = tmp_path / (name + ".out")
        r = subprocess.run([exe] + c["cmd"][1:] + ["-i", os.path.join(GAL, c["input"]), "-o", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert r.returncode == 0, (name, r.stderr[-500:])
        assert out.read_bytes() == open(os.path.join(GAL, c["expect"]), "rb").read(), name
