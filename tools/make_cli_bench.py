"""Synthetic coordinate-sorted BAM + VCF for timing the `demuxlet` binary's scan at a size the test
generators (per-read Python loops) cannot reach.   python tools/make_cli_bench.py OUTDIR N_READS N_SNPS N_SAMPLES N_BARCODES
Reads are 90M, mapq 60, CB:Z/UB:Z tagged, placed so that ~80 % cover one of the SNPs; genotypes are GT."""
import struct
import sys
import zlib

import numpy as np

out, n_reads, n_snps, n_samples, n_bc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rng = np.random.default_rng(5)
contigs = [("chr1", 120_000_000), ("chr2", 100_000_000), ("chr3", 80_000_000)]
BASES = "ACGT"

# ---- VCF
per = n_snps // len(contigs)
snp_pos = []
with open(f"{out}/bench.vcf", "w") as f:
    f.write("##fileformat=VCFv4.2\n")
    for name, length in contigs:
        f.write(f"##contig=<ID={name},length={length}>\n")
    f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
    f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{j}" for j in range(n_samples)) + "\n")
    gts = np.array(["0/0", "0/1", "1/1"])
    for name, length in contigs:
        pos = np.sort(rng.choice(np.arange(1000, length - 1000, 37), size=per, replace=False))
        snp_pos.append(pos)
        ref = rng.integers(0, 4, per)
        alt = (ref + 1 + rng.integers(0, 3, per)) % 4
        af = rng.uniform(0.1, 0.9, per)
        g = (rng.random((per, n_samples)) < af[:, None]).astype(int) + (rng.random((per, n_samples)) < af[:, None]).astype(int)
        for i in range(per):
            f.write(f"{name}\t{pos[i] + 1}\t.\t{BASES[ref[i]]}\t{BASES[alt[i]]}\t.\tPASS\t.\tGT\t" + "\t".join(gts[g[i]]) + "\n")

# ---- BAM
def bgzf_block(data):
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))

text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in contigs)
hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs))
for n, l in contigs:
    hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
bcs = [f"{''.join(BASES[i] for i in rng.integers(0, 4, 16))}-1" for _ in range(n_bc)]
nib = np.array([1, 2, 4, 8], dtype=np.uint8)
with open(f"{out}/bench.bam", "wb") as f:
    f.write(bgzf_block(hdr))
    buf = bytearray()
    per_c = n_reads // len(contigs)
    L = 90
    for tid, (name, length) in enumerate(contigs):
        sp = snp_pos[tid]
        on = rng.random(per_c) < 0.8
        start = np.where(on, sp[rng.integers(0, len(sp), per_c)] - rng.integers(0, L, per_c), rng.integers(0, length - 400, per_c))
        start = np.sort(np.maximum(start, 0))
        seq = rng.integers(0, 4, (per_c, L))
        packed = (nib[seq[:, 0::2]] << 4) | nib[seq[:, 1::2]]
        qual = rng.integers(20, 41, (per_c, L)).astype(np.uint8)
        bci = rng.integers(0, n_bc, per_c)
        umi = rng.integers(0, 4, (per_c, 10))
        for i in range(per_c):
            qn = b"r%d\0" % i
            tags = b"CBZ" + bcs[bci[i]].encode() + b"\0UBZ" + bytes(b"ACGT"[x] for x in umi[i]) + b"\0"
            core = struct.pack("<iiBBHHHiiii", tid, int(start[i]), len(qn), 60, 4680, 1, 0, L, -1, -1, 0)
            rec = core + qn + struct.pack("<I", (L << 4) | 0) + packed[i].tobytes() + qual[i].tobytes() + tags
            buf += struct.pack("<i", len(rec)) + rec
            if len(buf) >= 0xff00:
                f.write(bgzf_block(bytes(buf[:0xff00])))
                del buf[:0xff00]
    while buf:
        f.write(bgzf_block(bytes(buf[:0xff00])))
        del buf[:0xff00]
    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
print("wrote", out)
