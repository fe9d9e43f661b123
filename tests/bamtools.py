"""Test helper: writes small coordinate-sorted BAM files with their .bai, and FASTA files with their .fai (plain Python + zlib), so that
the native ingestion can be exercised on records the reference's example data set does not have (X / = operations, HP and MM / ML tags,
secondary / supplementary records, deep coverage)."""
import struct
import zlib

_SEQ = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _tags(tags):
    out = b""
    for k, v in tags.items():
        t, val = v
        out += k.encode() + t.encode()[:1]
        if t == "f":
            out += struct.pack("<f", val)
        elif t == "C":
            out += struct.pack("<B", val)
        elif t == "i":
            out += struct.pack("<i", val)
        elif t == "Z":
            out += val.encode() + b"\0"
        elif t == "BC":
            out = out[:-1] + b"B" + b"C" + struct.pack("<I", len(val)) + bytes(val)
        else:
            raise ValueError(t)
    return out


def encode_record(r):
    cig = [(n << 4) | _OPS[c] for c, n in r["cigar"]]
    ref_len = sum(n for c, n in r["cigar"] if c in "MDN=X")
    seq = r["seq"]
    packed = bytearray((len(seq) + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= _SEQ[ch] << (4 if i % 2 == 0 else 0)
    name = r["name"].encode() + b"\0"
    body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r.get("mapq", 60), reg2bin(r["pos"], r["pos"] + max(1, ref_len)), len(cig), r.get("flag", 0),
                       len(seq), -1, -1, 0)
    body += name + struct.pack("<%dI" % len(cig), *cig) + bytes(packed) + bytes(r.get("qual", [30] * len(seq))) + _tags(r.get("tags", {}))
    return struct.pack("<i", len(body)) + body, ref_len


def _bgzf_block(data):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def write_bam(path, refs, records, block=0x8000):
    """refs: [(name, length)]; records: dicts (name, tid, pos, cigar [(op char, len)], seq, optional mapq / flag / qual / tags), sorted."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs:
        head += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    stream = bytearray(head)
    spans = []  # (tid, beg, end, ustart, uend)
    for r in records:
        enc, ref_len = encode_record(r)
        spans.append((r["tid"], r["pos"], r["pos"] + max(1, ref_len), len(stream), len(stream) + len(enc)))
        stream += enc
    blocks, coffs, out = [], [], bytearray()
    for o in range(0, len(stream), block):
        coffs.append(len(out))
        out += _bgzf_block(bytes(stream[o:o + block]))
    eof_coff = len(out)
    out += _bgzf_block(b"")
    open(path, "wb").write(bytes(out))

    def voff(u):
        b = u // block
        if b >= len(coffs):
            return eof_coff << 16
        return (coffs[b] << 16) | (u % block)

    # .bai
    idx = b"BAI\1" + struct.pack("<i", len(refs))
    for tid in range(len(refs)):
        bins, linear = {}, {}
        for t, beg, end, us, ue in spans:
            if t != tid:
                continue
            bins.setdefault(reg2bin(beg, end), []).append((voff(us), voff(ue)))
            for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                linear[w] = min(linear.get(w, 1 << 63), voff(us))
        idx += struct.pack("<i", len(bins))
        for b, chunks in sorted(bins.items()):
            idx += struct.pack("<Ii", b, len(chunks)) + b"".join(struct.pack("<QQ", a, e) for a, e in chunks)
        n_intv = (max(linear) + 1) if linear else 0
        idx += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):
            last = linear.get(w, last)
            idx += struct.pack("<Q", last)
    open(path + ".bai", "wb").write(idx)


def write_fasta(path, contigs, width=60):
    """contigs: [(name, sequence)]"""
    fa, fai, off = "", "", 0
    for name, seq in contigs:
        hdr = ">%s\n" % name
        off += len(hdr)
        body = "\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + "\n"
        fai += "%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), off, width, width + 1)
        fa += hdr + body
        off += len(body)
    open(path, "w").write(fa)
    open(path + ".fai", "w").write(fai)


def read_bam_records(path):
    """All records of a BAM with their aux tags: dicts (name, tid, pos, mapq, flag, cigar [(op char, len)], seq, qual, tags {tag: value})."""
    import gzip
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", data, p)[0]
        refs.append((data[p + 4:p + 4 + ln - 1].decode(), struct.unpack_from("<i", data, p + 4 + ln)[0]))
        p += 8 + ln
    size = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}
    fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}
    out = []
    while p < len(data):
        bs = struct.unpack_from("<i", data, p)[0]
        rec = data[p + 4:p + 4 + bs]
        p += 4 + bs
        tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", rec, 0)
        q = 32
        name = rec[q:q + l_rn - 1].decode()
        q += l_rn
        cigar = [("MIDNSHP=X"[v & 0xF], v >> 4) for v in struct.unpack_from("<%dI" % n_cig, rec, q)]
        q += 4 * n_cig
        packed = rec[q:q + (l_seq + 1) // 2]
        q += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(packed[i >> 1] >> (4 if i % 2 == 0 else 0)) & 0xF] for i in range(l_seq))
        qual = list(rec[q:q + l_seq])
        q += l_seq
        tags = {}
        while q < len(rec):
            tag, ty = rec[q:q + 2].decode(), chr(rec[q + 2])
            q += 3
            if ty in size:
                tags[tag] = (ty, struct.unpack_from("<" + fmt.get(ty, "c"), rec, q)[0])
                q += size[ty]
            elif ty == "Z":
                e = rec.index(b"\0", q)
                tags[tag] = (ty, rec[q:e].decode())
                q = e + 1
            elif ty == "B":
                sub, n = chr(rec[q]), struct.unpack_from("<I", rec, q + 1)[0]
                tags[tag] = ("B" + sub, list(struct.unpack_from("<%d%s" % (n, fmt[sub]), rec, q + 5)))
                q += 5 + n * size[sub]
            else:
                raise ValueError(ty)
        out.append(dict(name=name, tid=tid, pos=pos, mapq=mapq, flag=flag, cigar=cigar, seq=seq, qual=qual, tags=tags, mtid=mtid, mpos=mpos))
    return text, refs, out
