"""`tracy_amd_cli basecall` (teal.h:24-117) runs without a GPU: the four output formats against the reference's own
abif.h where it is compilable (tsv = traceTxtOut, oracle/_ref) and against the Python restatement (json, fasta, fastq)."""
import os
import subprocess

import numpy as np
import pytest

import indigo_oracle as io
import sage_oracle as so
from test_host_and_abi import make_trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tracy_amd", "bin", "tracy_amd_cli")


@pytest.fixture(scope="module")
def trace_file(tmp_path_factory):
    from tracy_amd import build, hostlib
    build.build_host()
    if not os.path.exists(CLI):
        pytest.skip("tracy_amd_cli is not built (needs libtracy_hip.so: run __graft_entry__.build())")
    rng = np.random.default_rng(99)
    tr, pos = make_trace(rng, 160, het=0.25)
    tr = np.minimum(tr, 32000)
    p = str(tmp_path_factory.mktemp("bc") / "t.ab1")
    hostlib.write_abif(p, tr, pos, b"N" * 160, np.full(160, 20, np.uint8))
    return p, tr, pos


def run(args):
    return subprocess.run([CLI, "basecall"] + args, capture_output=True, text=True, timeout=120)


def test_basecall_formats(trace_file, tmp_path):
    from tracy_amd import hostlib
    path, tr, pos = trace_file
    pri, sec, con, bcpos, q = hostlib.basecall_qual(tr, pos, 0.33)
    out = str(tmp_path / "o")
    p = run(["-o", out + ".json", path])
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip().endswith("Done.")
    assert open(out + ".json").read() == "{\n" + io.trace_json_body(tr, bcpos.tolist(), q.tolist(), pri, sec) + "\n}\n"
    # tsv == traceTxtOut (pinned against the reference in tests/test_trace_io.py)
    assert run(["-f", "tsv", "-q", "5", "-u", "7", "-o", out + ".tsv", path]).returncode == 0
    assert hostlib.trace_txt(out + ".want", tr, pos, 0.33, 5, 7) == 0
    assert open(out + ".tsv").read() == open(out + ".want").read()
    # fasta / fastq with trims and the three sequences
    for otype, seq in (("primary", pri), ("secondary", sec), ("consensus", con)):
        assert run(["-f", "fasta", "-y", otype, "-q", "10", "-u", "20", "-o", out + ".fa", path]).returncode == 0
        assert open(out + ".fa").read() == ">%s\n%s\n" % (otype, seq[10:len(seq) - 20].decode())
    assert run(["--format=fastq", "--trimLeft", "3", "--trimRight", "4", "-o", out + ".fq", path]).returncode == 0
    want_q = "".join(chr(int(v) + 33) for v in q[3:len(q) - 4])
    assert open(out + ".fq").read() == "@primary\n%s\n+\n%s\n" % (pri[3:len(pri) - 4].decode(), want_q)
    # quality-based trimming
    assert run(["-f", "fasta", "-t", "3", "-o", out + ".t.fa", path]).returncode == 0
    tl, trr = so.trim_trace(3, sec, bcpos.tolist())
    assert open(out + ".t.fa").read() == ">primary\n%s\n" % pri[tl:len(pri) - trr].decode()


def test_basecall_errors(trace_file, tmp_path):
    path, _, _ = trace_file
    assert run([]).returncode == 255
    p = run([str(tmp_path / "missing.ab1")])
    assert p.returncode == 1 and "Input trace file is missing" in p.stderr
    junk = str(tmp_path / "junk.ab1")
    open(junk, "wb").write(b"not a trace at all")
    p = run([junk])
    assert p.returncode == 255 and "Unknown trace file type!" in p.stderr
    p = run(["-q", "100", "-u", "100", path])
    assert p.returncode == 255 and "larger than the trace" in p.stderr
