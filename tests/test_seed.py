"""Host k-mer seeding for indexed genomes (tracy_amd/host/seed.hpp; fmindex.h:173-326, BASELINE configs[3]) against a
brute-force restatement over the same text.  PARITY UNPINNED: the reference queries an sdsl-lite FM index (absent);
any exact index yields the same occurrence sets."""
import gzip
import os

import numpy as np
import pytest

import sage_oracle as so


def rand_dna(rng, n):
    return bytes(rng.choice(list(b"ACGT"), size=n).tolist()).decode()


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    from tracy_amd import hostlib
    rng = np.random.default_rng(123)
    rep = rand_dna(rng, 400)
    c1 = rand_dna(rng, 30000)
    c1 = c1[:5000] + rep + c1[5000:12000] + "N" * 300 + c1[12000:20000] + rep + c1[20000:]      # a repeat + an N run
    c2 = rand_dna(rng, 9000)
    c2 = c2[:3000] + rep + c2[3000:]
    c3 = rand_dna(rng, 2500)                                                                    # shorter than a window
    low = c2[:100].lower() + c2[100:]                                                           # lower case is upper-cased
    contigs = [("chrA", c1), ("chrB description text", c2), ("chrC", c3)]
    path = str(tmp_path_factory.mktemp("genome") / "toy.fa.gz")
    with gzip.open(path, "wt") as f:
        for (name, seq), body in zip(contigs, (c1, low, c3)):
            f.write(">%s\n" % name)
            for i in range(0, len(body), 60):
                f.write(body[i:i + 60] + "\n")
    g = hostlib.Genome(path, 15, 2)
    brute = so.BruteGenome([("chrA", c1), ("chrB", c2), ("chrC", c3)])
    return g, brute, path


def make_reads(rng, brute, n):
    reads = []
    text_contigs = brute.text.split("\n")[:-1]
    for i in range(n):
        ci = int(rng.integers(0, 3))
        seq = text_contigs[ci]
        L = int(rng.integers(150, 900))
        L = min(L, len(seq) - 1)
        start = int(rng.choice([0, len(seq) - L, rng.integers(0, len(seq) - L + 1)]))
        r = list(seq[start:start + L])
        for k in range(len(r)):
            u = rng.random()
            if u < 0.01:
                r[k] = "ACGT"[int(rng.integers(0, 4))]
            elif u < 0.02:
                r[k] = "N"
        r = "".join(r)
        if i % 2:
            r = so._revcomp_str(r)
        reads.append(r)
    reads.append(rand_dna(rng, 500))                    # unrelated: cannot be anchored
    reads.append("N" * 300)                             # no usable k-mer
    return reads


def test_counts_match_brute_force(genome):
    g, brute, _ = genome
    rng = np.random.default_rng(5)
    for _ in range(200):
        p = int(rng.integers(0, len(brute.text) - 15))
        pat = brute.text[p:p + 15]
        assert g.count(pat.encode()) == len(brute.locate(pat)), pat
    assert g.count(b"ACGTACGTACGTACG") == len(brute.locate("ACGTACGTACGTACG"))
    assert g.count(b"ACGTAC") == len(brute.locate("ACGTAC"))  # shorter than k: answered by a scan


# (trims of at least k - 1 on both sides: both strands in one pass, scanBothStrands; shorter ones: the two scans one after the other)
@pytest.mark.parametrize("trims,kmer_support,maxindel", [((50, 50), 3, 1000), ((10, 5), 3, 200), ((0, 0), 8, 1000), ((14, 40), 3, 500), ((60, 14), 2, 300),
                                                         ((13, 14), 3, 1000)])
def test_get_reference_slice(genome, trims, kmer_support, maxindel):
    g, brute, _ = genome
    rng = np.random.default_rng(77 + trims[0])
    reads = make_reads(rng, brute, 40)
    got = g.seed([r.encode() for r in reads], trims[0], trims[1], kmer_support, maxindel, 2)
    anchored = 0
    for i, r in enumerate(reads):
        want = so.get_reference_slice(brute, r, trims[0], trims[1], 15, kmer_support, maxindel)
        if want is None:
            assert got["status"][i] == 0, i
            continue
        anchored += 1
        assert got["status"][i] == 1, i
        assert bool(got["forward"][i]) == want["forward"] and int(got["kmersupport"][i]) == want["kmersupport"], i
        assert int(got["pos"][i]) == want["pos"] and int(got["contig"][i]) == want["contig"], i
        assert got["slices"][i].decode() == want["refslice"], i
    # (with trimRight < kmer the tail patterns get shorter than k and even an unrelated read can collect votes --
    # the reference behaves the same way, fmindex.h:211-214)
    assert anchored >= 30 and got["status"][-1] == 0
    if trims[1] >= 15:
        assert got["status"][-2] == 0


def test_reads_shorter_than_a_trim(genome):
    """a trim longer than the read wraps the reference's loop bound (fmindex.h:211: size() - trimRight in size_t): windows [trimLeft, size)
    still vote on that strand -- the one-pass form of both strands must leave such reads to the two scans"""
    g, brute, _ = genome
    rng = np.random.default_rng(4711)
    reads = []
    for _ in range(20):
        p = int(rng.integers(0, len(brute.text) - 200))
        r = brute.text[p:p + int(rng.integers(90, 130))]
        reads.append(r if rng.random() < 0.5 else so.revcomp(r.encode()).decode())
    voted = 0
    for trims in ((20, 150), (150, 20), (14, 140)):
        got = g.seed([r.encode() for r in reads], trims[0], trims[1], 3, 1000, 2)
        for i, r in enumerate(reads):
            want = so.get_reference_slice(brute, r, trims[0], trims[1], 15, 3, 1000)
            assert (got["status"][i] == 1) == (want is not None), (trims, i)
            if want is not None:
                voted += 1
                assert bool(got["forward"][i]) == want["forward"] and int(got["kmersupport"][i]) == want["kmersupport"], (trims, i)
                assert int(got["pos"][i]) == want["pos"] and got["slices"][i].decode() == want["refslice"], (trims, i)
    assert voted >= 20


def test_repeat_reads_need_the_second_pass(genome):
    """a read lying inside the 3-copy repeat has no unique 15-mers: the first pass fails, the multi-hit pass ties"""
    g, brute, _ = genome
    rep_start = brute.text.find(brute.text[5000:5400])
    read = brute.text[rep_start + 20:rep_start + 380]
    want = so.get_reference_slice(brute, read, 10, 10)
    got = g.seed([read.encode()], 10, 10)
    assert (got["status"][0] == 1) == (want is not None)
    if want is not None:
        assert got["slices"][0].decode() == want["refslice"] and int(got["pos"][0]) == want["pos"]


def test_index_file_equals_the_in_memory_build(genome, tmp_path):
    """`tracy index` (index.h:79-124): the table written to a file and mapped back answers every query like the table built in
    memory from the FASTA -- counts (table and scan paths), getReferenceSlice on a batch of reads -- and a corrupt / truncated / wrong-k
    file is refused"""
    import subprocess
    from tracy_amd import hostlib
    g, brute, path = genome
    idx = str(tmp_path / "toy.tidx")
    g.save(idx)
    m = hostlib.Genome(idx, 15, 2)
    rng = np.random.default_rng(9)
    for _ in range(100):
        p = int(rng.integers(0, len(brute.text) - 15))
        pat = brute.text[p:p + 15].encode()
        assert m.count(pat) == g.count(pat)
    assert m.count(b"ACGTAC") == g.count(b"ACGTAC")
    reads = [r.encode() for r in make_reads(rng, brute, 30)]
    a, b = g.seed(reads, 50, 50, 3, 1000, 2), m.seed(reads, 50, 50, 3, 1000, 2)
    for k in ("status", "forward", "kmersupport", "pos", "contig"):
        assert np.array_equal(a[k], b[k]), k
    assert a["slices"] == b["slices"]
    # the command line writes the same file
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tracy_amd", "bin", "tracy_amd_cli")
    out = str(tmp_path / "cli.tidx")
    p = subprocess.run([cli, "index", "-o", out, path], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert open(out, "rb").read() == open(idx, "rb").read()
    # refused: another k, a truncated file, junk behind the magic
    with pytest.raises(IOError):
        hostlib.Genome(idx, 11)
    blob = open(idx, "rb").read()
    import struct
    k_bits, tbytes, nt, nc, nmb = struct.unpack_from("<5Q", blob, 8)
    al = lambda x: (x + 7) & ~7
    at_len = 56 + al(tbytes) + al(nmb)
    at_st = at_len + al(4 * nc)
    at_bkt = at_st + al(8 * nc)

    def patched(off, fmt, *v):
        b = bytearray(blob)
        struct.pack_into(fmt, b, off, *v)
        return bytes(b)
    wrapped = [patched(8 + 8 * 2, "<Q", (1 << 61) + 1),  # table entries: 8 * count wraps 64 bits
               patched(8 + 8 * 3, "<Q", (1 << 62) + nc),  # contigs: 4 * count wraps
               patched(at_bkt + 8 * 5, "<Q", nt + 7),       # directory not a prefix sum of the table
               patched(at_bkt, "<Q", 3),
               patched(at_st, "<Q", tbytes + 1),            # a contig outside the text
               patched(at_len, "<I", 0xfffffff0)]
    for bad in [blob[:len(blob) // 2], blob[:8] + b"\xff" * 200, blob[:60]] + wrapped:
        open(str(tmp_path / "bad.tidx"), "wb").write(bad)
        with pytest.raises(IOError):
            hostlib.Genome(str(tmp_path / "bad.tidx"), 15)
    m.close()


def test_even_k_and_palindromes(tmp_path):
    """k = 12: a k-mer can be its own reverse complement (the run of the table then holds one part only, and both strands of a read vote
    from it) -- planted palindromes, reads across them on both strands, both trims regimes"""
    from tracy_amd import hostlib
    rng = np.random.default_rng(12)
    pal = lambda h: h + so._revcomp_str(h)  # noqa: E731
    body = rand_dna(rng, 20000)
    for at in range(500, 19000, 900):
        body = body[:at] + pal(rand_dna(rng, 6)) + body[at + 12:]
    path = str(tmp_path / "pal.fa")
    with open(path, "w") as f:
        f.write(">chrP\n" + body + "\n")
    g = hostlib.Genome(path, 12, 2)
    brute = so.BruteGenome([("chrP", body)])
    reads = []
    for i in range(40):
        st = int(rng.integers(0, len(body) - 400))
        r = body[st:st + int(rng.integers(120, 380))]
        reads.append(so._revcomp_str(r) if i % 2 else r)
    for trims in ((20, 20), (11, 30), (3, 2)):
        got = g.seed([r.encode() for r in reads], trims[0], trims[1], 3, 300, 2)  # (k = the table's: 12)
        for i, r in enumerate(reads):
            want = so.get_reference_slice(brute, r, trims[0], trims[1], 12, 3, 300)
            assert (got["status"][i] == 1) == (want is not None), (trims, i)
            if want is not None:
                assert bool(got["forward"][i]) == want["forward"] and int(got["kmersupport"][i]) == want["kmersupport"], (trims, i)
                assert int(got["pos"][i]) == want["pos"] and got["slices"][i].decode() == want["refslice"], (trims, i)
    g.close()


def test_packed_batches_and_reused_buffers(genome):
    """seed_packed (the batch packed once, result buffers reused from call to call -- what bench.py's seeding leg does) returns what
    seed() returns, also when the buffers held another batch's results before"""
    from tracy_amd import hostlib
    g, brute, _ = genome
    rng = np.random.default_rng(31)
    ra, rb = make_reads(rng, brute, 24), make_reads(rng, brute, 24)
    ra, rb = [r.encode() for r in ra], [r.encode() for r in rb]
    cap = max(len(r) for r in ra + rb)
    rb = [r.ljust(cap, b"A") if i == 0 else r for i, r in enumerate(rb)]  # (same longest read: the buffers of the first call fit the second)
    ra = [r.ljust(cap, b"A") if i == 0 else r for i, r in enumerate(ra)]
    want_a, want_b = g.seed(ra, 50, 50, 3, 400, 2, raw=True), g.seed(rb, 50, 50, 3, 400, 2, raw=True)
    out = g.seed_packed(hostlib.Genome.pack_consensus(ra), 50, 50, 3, 400, 2)
    for k in ("status", "forward", "kmersupport", "pos", "contig", "slice_len"):
        assert np.array_equal(out[k], want_a[k]), k
    out2 = g.seed_packed(hostlib.Genome.pack_consensus(rb), 50, 50, 3, 400, 2, out=out)
    assert out2 is out
    for k in ("status", "forward", "pos", "contig", "slice_len"):
        assert np.array_equal(out2[k], want_b[k]), k
    for i in range(len(rb)):
        if want_b["status"][i] == 1:
            n = int(want_b["slice_len"][i])
            assert out2["slices_2d"][i, :n].tobytes() == want_b["slices_2d"][i, :n].tobytes(), i
