"""A minimal BCF2.2 reader (TEST INFRASTRUCTURE): BGZF blocks -> header text + records decoded to the columns of a VCF line.
Written from the SAM/VCF specification (sections BGZF and BCF2), independently of tracy_amd/host/bcf_out.hpp."""
import struct
import zlib


def bgzf_decompress(data):
    out, at, blocks = bytearray(), 0, 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04", "gzip member with an extra field"
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        extra = data[at + 12:at + 12 + xlen]
        assert extra[:4] == b"BC\x02\x00", "BGZF extra subfield"
        bsize = struct.unpack_from("<H", extra, 4)[0] + 1
        payload = data[at + 12 + xlen:at + bsize - 8]
        crc, isize = struct.unpack_from("<II", data, at + bsize - 8)
        raw = zlib.decompress(payload, -15)
        assert len(raw) == isize and zlib.crc32(raw) == crc
        out += raw
        at += bsize
        blocks += 1
    assert len(raw) == 0, "the last block is the empty end-of-file marker"
    return bytes(out), blocks


class Cur:
    def __init__(self, b):
        self.b, self.at = b, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.at)
        self.at += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def typed(self):
        d = self.take("B")
        n, t = d >> 4, d & 15
        if n == 15:
            n = self.typed_int()
        return n, t

    def typed_int(self):
        n, t = self.typed()
        assert n == 1 and t in (1, 2, 3)
        return self.take({1: "b", 2: "h", 3: "i"}[t])

    def values(self):
        n, t = self.typed()
        if t == 7:
            s = self.b[self.at:self.at + n].decode()
            self.at += n
            return s
        if t == 0:
            return None
        return [self.take({1: "b", 2: "h", 3: "i", 5: "f"}[t]) for _ in range(n)]


def read_bcf(path):
    raw, blocks = bgzf_decompress(open(path, "rb").read())
    assert raw[:5] == b"BCF\x02\x02"
    l_text = struct.unpack_from("<I", raw, 5)[0]
    text = raw[9:9 + l_text]
    assert text[-1] == 0
    header = text[:-1].decode()
    # dictionaries: FILTER / INFO / FORMAT ids in order of appearance (PASS = 0), contigs in order
    ids, contigs = ["PASS"], []
    for ln in header.split("\n"):
        for kind in ("##FILTER=<ID=", "##INFO=<ID=", "##FORMAT=<ID="):
            if ln.startswith(kind):
                name = ln[len(kind):].split(",")[0].rstrip(">")
                if name not in ids:
                    ids.append(name)
        if ln.startswith("##contig=<ID="):
            contigs.append(ln[len("##contig=<ID="):].split(",")[0].rstrip(">"))
    c = Cur(raw)
    c.at = 9 + l_text
    recs = []
    while c.at < len(raw):
        l_shared, l_indiv = c.take("II")
        end_shared = c.at + l_shared
        chrom, pos, rlen = c.take("iii")
        qual = c.take("f")
        n_allele_info, n_fmt_sample = c.take("II")
        n_allele, n_info, n_fmt, n_sample = n_allele_info >> 16, n_allele_info & 0xffff, n_fmt_sample >> 24, n_fmt_sample & 0xffffff
        vid = c.values()
        alleles = [c.values() for _ in range(n_allele)]
        flt = [ids[i] for i in (c.values() or [])]
        info = {}
        for _ in range(n_info):
            key = ids[c.typed_int()]
            v = c.values()
            info[key] = v if isinstance(v, str) else v[0]
        assert c.at == end_shared
        end_indiv = c.at + l_indiv
        fmt = {}
        for _ in range(n_fmt):
            key = ids[c.typed_int()]
            n, t = c.typed()
            fmt[key] = [[c.take({1: "b", 2: "h", 3: "i"}[t]) for _ in range(n)] for _ in range(n_sample)]
        assert c.at == end_indiv
        gt = fmt["GT"][0]
        gts = "/".join("." if a == 0 else str((a >> 1) - 1) for a in gt)
        assert rlen == len(alleles[0])
        recs.append({"CHROM": contigs[chrom], "POS": pos + 1, "ID": vid, "REF": alleles[0], "ALT": ",".join(alleles[1:]), "QUAL": qual,
                     "FILTER": ";".join(flt), "INFO": info, "GT": gts, "GQ": fmt["GQ"][0][0]})
    return header, recs, blocks


# ---- CSI index (SAM/VCF specification, section "CSI index format"), read independently of tracy_amd/host/bcf_out.hpp ----
def bgzf_blocks(data):
    """[(file offset of the block, its uncompressed bytes)]"""
    out, at = [], 0
    while at < len(data):
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        bsize = struct.unpack_from("<H", data, at + 16)[0] + 1
        raw = zlib.decompress(data[at + 12 + xlen:at + bsize - 8], -15)
        out.append((at, raw))
        at += bsize
    return out


def read_csi(path):
    raw, _ = bgzf_decompress(open(path, "rb").read())
    assert raw[:4] == b"CSI\x01"
    min_shift, depth, l_aux = struct.unpack_from("<iii", raw, 4)
    at = 16 + l_aux
    n_ref = struct.unpack_from("<i", raw, at)[0]
    at += 4
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", raw, at)[0]
        at += 4
        bins = {}
        for _ in range(n_bin):
            b, loff, n_chunk = struct.unpack_from("<IQi", raw, at)
            at += 16
            chunks = [struct.unpack_from("<QQ", raw, at + 16 * k) for k in range(n_chunk)]
            at += 16 * n_chunk
            bins[b] = (loff, chunks)
        refs.append(bins)
    n_no_coor = struct.unpack_from("<Q", raw, at)[0] if at + 8 <= len(raw) else None
    return {"min_shift": min_shift, "depth": depth, "refs": refs, "n_no_coor": n_no_coor}


def reg2bins(beg, end, min_shift, depth):
    """every bin that may hold a record overlapping [beg, end) (the specification's reg2bins)"""
    out, t, s = [], 0, min_shift + depth * 3
    end -= 1
    for l in range(depth + 1):
        out.extend(range(t + (beg >> s), t + (end >> s) + 1))
        t += 1 << (l * 3)
        s -= 3
    return out


def csi_query(bcf_path, csi, rid, beg, end):
    """records of contig `rid` overlapping [beg, end) found THROUGH the index: (pos0, rlen) of every record in the chunks of the bins
    the region maps to, filtered by overlap -- read at the chunks' virtual offsets"""
    data = open(bcf_path, "rb").read()
    blocks = bgzf_blocks(data)
    start_of = {off: sum(len(r) for _, r in blocks[:i]) for i, (off, _) in enumerate(blocks)}
    flat = b"".join(r for _, r in blocks)

    def upos(v):
        return start_of[v >> 16] + (v & 0xffff)
    meta = ((1 << ((csi["depth"] + 1) * 3)) - 1) // 7 + 1
    found = []
    for b in reg2bins(beg, end, csi["min_shift"], csi["depth"]):
        if b == meta or b not in csi["refs"][rid]:
            continue
        for cb, ce in csi["refs"][rid][b][1]:
            at, stop = upos(cb), upos(ce)
            while at < stop:
                l_shared, l_indiv = struct.unpack_from("<II", flat, at)
                chrom, pos, rlen = struct.unpack_from("<iii", flat, at + 8)
                if chrom == rid and pos < end and pos + max(rlen, 1) > beg:
                    found.append((pos, rlen))
                at += 8 + l_shared + l_indiv
            assert at == stop, "a chunk ends on a record boundary"
    return sorted(set(found))
