"""Test helper: synthetic coordinate-sorted SAM/BAM + VCF text for the CLI's scan, and an
independent Python restatement of the reference's lock-step scan (cmd_cram_demuxlet.cpp:195-338, hts_utils.cpp:279-359,
sam_filtered_reader.cpp:284-296, bcf_filtered_reader.cpp:498-574,649-669) used as the expectation."""
import gzip
import struct

import numpy as np

BASES = "ACGT"


def make_vcf(rng, contigs, n_per_contig, samples, path, with_noise=True, contig_lines=True):
    """Returns the list of records written (dicts) in file order."""
    recs = []
    lines = ["##fileformat=VCFv4.2"]
    for name, length in contigs:
        if contig_lines:                           # without them a VCF teaches its contigs record by record (bcf_filtered_reader / htslib)
            lines.append(f"##contig=<ID={name},length={length}>")
    lines += ['##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">', '##FORMAT=<ID=PL,Number=G,Type=Integer,Description="PL">',
              '##FORMAT=<ID=GP,Number=G,Type=Float,Description="GP">']
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples))
    for name, length in contigs:
        poss = np.sort(rng.choice(np.arange(100, length - 100), size=n_per_contig, replace=False))
        for pos in poss:
            ref = BASES[rng.integers(0, 4)]
            alt = BASES[(BASES.index(ref) + 1 + rng.integers(0, 3)) % 4]
            kind = rng.random() if with_noise else 1.0
            if kind < 0.04:
                alt = alt + "," + BASES[(BASES.index(ref) + 2) % 4] if BASES[(BASES.index(ref) + 2) % 4] != alt else alt + ",N"   # multi-allelic: filtered (:534)
            elif kind < 0.08:
                ref = ref + "AT"                                                                                                    # deletion: only warned about (:215-225)
            af = rng.uniform(0.1, 0.9)
            fields = []
            for _ in samples:
                if rng.random() < (0.15 if with_noise else 0.0):
                    fields.append("./.:.:.")
                    continue
                a = sorted(int(x) for x in (rng.random(2) < af))
                if rng.random() < 0.03 and with_noise:
                    gt = f"{a[0]}/."
                else:
                    gt = f"{a[0]}{'/' if rng.random() < 0.7 else '|'}{a[1]}"
                g = a[0] + a[1]
                pl = [int(x) for x in rng.integers(10, 200, size=3)]; pl[g] = 0
                gp = rng.uniform(0, 0.05, size=3); gp[g] = 0.9
                fields.append(f"{gt}:{','.join(map(str, pl))}:{','.join(f'{x:.4f}' for x in gp)}")
            lines.append(f"{name}\t{pos + 1}\t.\t{ref}\t{alt}\t50\tPASS\t.\tGT:PL:GP\t" + "\t".join(fields))
            recs.append(dict(chrom=name, pos=int(pos), ref=ref, alt=alt, fields=fields))
    text = "\n".join(lines) + "\n"
    if str(path).endswith(".gz"):
        with gzip.open(path, "wt") as f:
            f.write(text)
    else:
        open(path, "w").write(text)
    return recs


def make_reads(rng, contigs, recs, n_reads, barcodes, path_sam, path_bam=None):
    """Coordinate-sorted reads, many of them placed over variants; returns the parsed read dicts (after writing)."""
    import bisect
    by_chrom = {}
    for r in recs:
        by_chrom.setdefault(r["chrom"], []).append(r)
    pos_of = {c: [v["pos"] for v in vs] for c, vs in by_chrom.items()}        # (file order = ascending position inside a contig)
    assert all(p == sorted(p) for p in pos_of.values())
    reads = []
    tid_of = {c[0]: i for i, c in enumerate(contigs)}
    for _ in range(n_reads):
        cname, clen = contigs[rng.integers(0, len(contigs))]
        if rng.random() < 0.8 and by_chrom.get(cname):
            v = by_chrom[cname][rng.integers(0, len(by_chrom[cname]))]
            start = max(0, v["pos"] - int(rng.integers(0, 80)))
        else:
            start = int(rng.integers(0, clen - 400))
        shape = rng.integers(0, 6)
        cig = [[("M", 90)], [("M", 30), ("N", 200), ("M", 60)], [("S", 5), ("M", 85)], [("M", 40), ("I", 2), ("M", 48)],
               [("M", 40), ("D", 3), ("M", 50)], [("M", 20), ("S", 10)]][shape]
        qlen = sum(n for op, n in cig if op in "MIS")
        seq = "".join(BASES[i] for i in rng.integers(0, 4, size=qlen))
        # make the read carry REF or ALT at covered SNPs (so alleles 0/1 dominate)
        seq = list(seq)
        cpos, rpos = start, 0
        for op, n in cig:
            if op == "M":
                vs, ps = by_chrom.get(cname, []), pos_of.get(cname, [])
                for v in vs[bisect.bisect_left(ps, cpos):bisect.bisect_left(ps, cpos + n)]:     # the variants under this block, in file order
                    if rng.random() < 0.9:
                        seq[rpos + v["pos"] - cpos] = (v["ref"][0] if rng.random() < 0.5 else v["alt"][0])
                cpos += n; rpos += n
            elif op in "DN":
                cpos += n
            else:
                rpos += n
        if rng.random() < 0.02:
            seq[rng.integers(0, qlen)] = "N"
        qual = "".join(chr(33 + int(q)) for q in rng.integers(2, 42, size=qlen))
        flag = int(rng.choice([0, 16, 0, 16, 0x400, 0x100, 0x4], p=[0.4, 0.4, 0.05, 0.05, 0.04, 0.03, 0.03]))
        mapq = int(rng.choice([60, 255, 30, 10, 0], p=[0.5, 0.2, 0.15, 0.1, 0.05]))
        tags = []
        if rng.random() < 0.97:
            tags.append("CB:Z:" + barcodes[rng.integers(0, len(barcodes))])
        if rng.random() < 0.97:
            tags.append("UB:Z:" + "".join(BASES[i] for i in rng.integers(0, 4, size=3)))
        tags.append("NH:i:1")
        reads.append(dict(qname=f"r{len(reads)}", flag=flag, chrom=cname, tid=tid_of[cname], pos=start, mapq=mapq, cigar=cig,
                          seq="".join(seq), qual=qual, tags=tags))
    reads.sort(key=lambda r: (r["tid"], r["pos"]))
    with open(path_sam, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n")
        for name, length in contigs:
            f.write(f"@SQ\tSN:{name}\tLN:{length}\n")
        for r in reads:
            cg = "".join(f"{n}{op}" for op, n in r["cigar"])
            f.write("\t".join([r["qname"], str(r["flag"]), r["chrom"], str(r["pos"] + 1), str(r["mapq"]), cg, "*", "0", "0", r["seq"], r["qual"]] + r["tags"]) + "\n")
    if path_bam:
        write_bam(contigs, reads, path_bam)
    return reads


def bgzf_compress(data: bytes, block: int = 0xff00) -> bytes:
    """BGZF framing (SAM spec 4.1): independent gzip members of <= 64 KiB with the member size in a 'BC' extra field,
    closed by the 28-byte empty EOF block — what samtools/bgzip write."""
    import zlib
    out = bytearray()
    for o in list(range(0, len(data), block)) + [None]:
        chunk = b"" if o is None else data[o:o + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out += struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bsize)
        out += body + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    return bytes(out)


def write_bam(contigs, reads, path, bgzf=True):
    """Minimal BAM writer: BGZF blocks as samtools writes them, or (bgzf=False) one ordinary gzip member."""
    out = bytearray(b"BAM\x01")
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in contigs)
    out += struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs))
    for n, l in contigs:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    for r in reads:
        name = r["qname"].encode() + b"\0"
        cig = b"".join(struct.pack("<I", (n << 4) | "MIDNSHP=X".index(op)) for op, n in r["cigar"])
        l = len(r["seq"])
        sq = bytearray((l + 1) // 2)
        for i, ch in enumerate(r["seq"]):
            sq[i // 2] |= code[ch] << (0 if i & 1 else 4)
        ql = bytes(ord(c) - 33 for c in r["qual"])
        aux = b""
        for t in r["tags"]:
            k, ty, v = t.split(":", 2)
            aux += k.encode() + (b"Z" + v.encode() + b"\0" if ty == "Z" else b"C" + bytes([int(v)]))
        body = struct.pack("<iiBBHHHIiii", r["tid"], r["pos"], len(name), r["mapq"], 4680, len(r["cigar"]), r["flag"], l, -1, -1, 0) + name + cig + bytes(sq) + ql + aux
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        f.write(bgzf_compress(bytes(out)) if bgzf else gzip.compress(bytes(out)))


# ---- independent restatement of the scan --------------------------------------------------------------------------------
def base_at(read, pos):
    cpos, rpos, rlen = read["pos"], 0, len(read["seq"])
    hit = False
    for op, n in read["cigar"]:
        if op == "M":
            if cpos <= pos <= cpos + n - 1:
                rpos += pos - cpos; hit = True; break
            cpos += n; rpos += n
        elif op in "DN":
            if cpos <= pos <= cpos + n - 1:
                rpos = -1; break
            cpos += n
        elif op in "SI":
            rpos += n
    if rpos < 0 or rpos >= rlen:
        return None
    return read["seq"][rpos], ord(read["qual"][rpos]) - 33, rpos


def endpos(read):
    rl = sum(n for op, n in read["cigar"] if op in "MDN=X") if not (read["flag"] & 4) else 0
    return read["pos"] + (rl if rl > 0 else 1)


def vcf_pass(rec, sm_cols, min_mac=1, min_callrate=0.5):
    alts = rec["alt"].split(",")
    if 1 + len(alts) > 2:
        return None
    an, ac0, ac = 0, 0, 0
    alleles = []
    for c in sm_cols:
        gt = rec["fields"][c].split(":")[0]
        parts = gt.replace("|", "/").split("/")
        a = [(-1 if p in (".", "") else int(p)) for p in parts] + [-1]
        a = a[:2]
        alleles.append(a)
        for x in a:
            if x >= 0:
                an += 1
                if x > 0: ac += 1
    if min_callrate > an / (2.0 * len(sm_cols)):
        return None
    if ac < min_mac or an - ac < min_mac:
        return None
    return alleles


def scan(reads, recs, contigs, samples, sm_ids=None, min_mq=20, excl_flag=0x0f04, min_bq=13, cap_bq=40, min_td=0, group=None):
    """Returns (snps kept [rec index], events [(barcode, snp_id or -1, umi, allele, bq, newread)], per-snp allele matrix)."""
    rid_of = {c[0]: i for i, c in enumerate(contigs)}
    sm_cols = list(range(len(samples))) if not sm_ids else [samples.index(s) for s in sorted(set(sm_ids))]
    passing = [(i, vcf_pass(r, sm_cols)) for i, r in enumerate(recs)]
    passing = [(i, a) for i, a in passing if a is not None]
    snps, gts = [], []
    it = iter(passing)

    def read_variant():
        nonlocal snps
        try:
            i, a = next(it)
        except StopIteration:
            return False
        snps.append(i); gts.append(a)
        return True
    assert read_variant()
    ibeg, nbuf, veof = 0, 1, False
    events = []
    for r in reads:
        if r["mapq"] < min_mq or (excl_flag & r["flag"]):
            continue
        rid = rid_of[r["chrom"]]
        n_rm = 0
        for i in range(nbuf):
            v = recs[snps[ibeg + i]]
            vr = rid_of[v["chrom"]]
            if vr < rid or (vr == rid and v["pos"] + len(v["ref"]) < r["pos"]):
                n_rm += 1
            else:
                break
        nbuf -= n_rm; ibeg += n_rm
        ep = endpos(r)
        while not veof:
            c = recs[snps[-1]]
            cr = rid_of[c["chrom"]]
            if cr < rid or (cr == rid and c["pos"] < ep):
                if read_variant(): nbuf += 1
                else: veof = True
            else:
                break
        cb = next((t[5:] for t in r["tags"] if t.startswith("CB:Z:")), ".")
        if group and cb not in group:
            continue
        ub = next((t[5:] for t in r["tags"] if t.startswith("UB:Z:")), ".")
        first = True
        for i in range(ibeg, ibeg + nbuf):
            v = recs[snps[i]]
            b = base_at(r, v["pos"])
            if b is None or b[0] == "N":
                continue
            base, q, rpos = b
            if q < min_bq or rpos < min_td - 1 or rpos + min_td > len(r["seq"]):
                continue
            al = 0 if base == v["ref"][0] else (1 if base == v["alt"].split(",")[0][0] else 2)
            events.append((cb, i, ub, al, min(q, cap_bq), 1 if first else 0))
            first = False
        if first:
            events.append((cb, -1, ".", 0, 0, 1))
    return snps, events, gts, sm_cols


def vcf_text_to_bcf(vcf_text: str, path, block: int = 0xff00, int_type: int = 2, idx_attrs: bool = False):
    """Writes VCF text as BCF2.2 (VCF spec 6.3: 'BCF\\2\\2', l_text, header text, then per record l_shared | l_indiv | shared block |
    per-sample block; typed values; GT as (allele + 1) << 1 | phased with 0 = missing allele and the end-of-vector code padding
    haploid calls) inside BGZF — what `bcftools view -Ob` writes.  Test helper for the binary's BCF reader: FILTER is written,
    INFO dropped (the scan reads neither), integer FORMAT vectors use int8 (GT) and `int_type` (1/2/3) for the rest."""
    lines = vcf_text.split("\n")
    header = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if l and not l.startswith("#")]
    dict_ids, contigs = ["PASS"], []
    out_header = []
    fmt_type = {}
    for l in header:
        if l.startswith("##FORMAT=<"):
            fmt_type[l.split("ID=")[1].split(",")[0].rstrip(">")] = l.split("Type=")[1].split(",")[0].rstrip(">")
        if l.startswith("##contig=<"):
            cid = l.split("ID=")[1].split(",")[0].rstrip(">")
            if idx_attrs:
                l = l[:-1] + f",IDX={len(contigs)}>"
            contigs.append(cid)
        elif l.startswith(("##FILTER=<", "##INFO=<", "##FORMAT=<")):
            fid = l.split("ID=")[1].split(",")[0].rstrip(">")
            if fid not in dict_ids:
                if idx_attrs:
                    l = l[:-1] + f",IDX={len(dict_ids)}>"
                dict_ids.append(fid)
        out_header.append(l)
    text = ("\n".join(out_header) + "\n").encode() + b"\0"
    n_sample = len(header[-1].split("\t")) - 9
    INT_FMT = {1: ("<b", -128, -127), 2: ("<h", -32768, -32767), 3: ("<i", -2147483648, -2147483647)}

    def desc(n, t):
        return bytes([(n << 4) | t]) if n < 15 else bytes([0xF0 | t, 0x11, n]) if n < 128 else bytes([0xF0 | t, 0x12]) + struct.pack("<h", n)

    def tstr(sv):
        b = sv.encode()
        return desc(len(b), 7) + b

    raw = bytearray(b"BCF\x02\x02" + struct.pack("<I", len(text)) + text)
    for l in body:
        f = l.split("\t")
        alts = [] if f[4] == "." else f[4].split(",")
        keys = f[8].split(":")
        shared = struct.pack("<iii", contigs.index(f[0]), int(f[1]) - 1, len(f[3]))
        shared += struct.pack("<I", 0x7F800001) if f[5] == "." else struct.pack("<f", float(f[5]))
        shared += struct.pack("<II", ((1 + len(alts)) << 16) | 0, (len(keys) << 24) | n_sample)
        shared += (b"\x07" if f[2] == "." else tstr(f[2])) + tstr(f[3]) + b"".join(tstr(a) for a in alts)
        shared += b"\x00" if f[6] == "." else b"\x11" + bytes([dict_ids.index(f[6])])
        indiv = bytearray()
        sf = [x.split(":") for x in f[9:]]
        for ki, key in enumerate(keys):
            vals = [s[ki] if ki < len(s) else "." for s in sf]
            indiv += b"\x11" + bytes([dict_ids.index(key)])
            if key == "GT":
                rows = []
                for v in vals:
                    parts = v.replace("|", "/").split("/")
                    phased = "|" in v
                    row = [0 if a in (".", "") else ((int(a) + 1) << 1) | (1 if phased and h > 0 else 0) for h, a in enumerate(parts)]
                    rows.append(row)
                n = max(2, max(len(r) for r in rows))
                indiv += desc(n, 1)
                for r in rows:
                    indiv += bytes([x & 0xFF for x in r] + [0x81] * (n - len(r)))
            elif fmt_type.get(key) in ("String", "Character"):
                bs = [b"." if v == "" else v.encode() for v in vals]
                n = max(len(b) for b in bs)
                indiv += desc(n, 7)
                for b in bs:
                    indiv += b + b"\0" * (n - len(b))
            elif key == "GP" or fmt_type.get(key) == "Float":
                rows = [[] if v == "." else v.split(",") for v in vals]
                n = max(1, max(len(r) for r in rows))
                indiv += desc(n, 5)
                for r in rows:
                    if not r:
                        indiv += struct.pack("<I", 0x7F800001) + struct.pack("<I", 0x7F800002) * (n - 1)
                        continue
                    for x in r:
                        indiv += struct.pack("<I", 0x7F800001) if x == "." else struct.pack("<f", float(x))
                    indiv += struct.pack("<I", 0x7F800002) * (n - len(r))
            else:                        # integer vectors (PL, ...)
                rows = [[] if v == "." else v.split(",") for v in vals]
                big = max([abs(int(x)) for r in rows for x in r if x != "."] + [0])
                it = max(int_type, 1 if big <= 120 else 2 if big <= 32000 else 3)      # like bcftools: the smallest type that fits
                fmt, miss, end = INT_FMT[it]
                n = max(1, max(len(r) for r in rows))
                indiv += desc(n, it)
                for r in rows:
                    if not r:
                        indiv += struct.pack(fmt, miss) + struct.pack(fmt, end) * (n - 1)
                        continue
                    for x in r:
                        indiv += struct.pack(fmt, miss if x == "." else int(x))
                    indiv += struct.pack(fmt, end) * (n - len(r))
        raw += struct.pack("<II", len(shared), len(indiv)) + shared + indiv
    open(path, "wb").write(bgzf_compress(bytes(raw), block=block))
