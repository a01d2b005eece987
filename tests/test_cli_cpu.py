"""CPU: the `demuxlet` front end (rows f1-f4) up to the pileup — option parsing, SAM/BAM/VCF readers, variant filter,
sample selection, CIGAR walk, base filters — via `--pileup-only`, against an independent Python restatement of the
reference's scan (tests/sam_vcf_synth.py).  PARITY-UNPINNED by the reference (its scan needs htslib)."""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import sam_vcf_synth as sv

ROOT = Path(__file__).resolve().parents[1]
CLI = ROOT / "demuxlet_amd" / "demuxlet"
CONTIGS = [("1", 30000), ("2", 20000), ("X", 15000)]
SAMPLES = ["smC", "smA", "smB", "smD"]


@pytest.fixture(scope="module")
def cli():
    from demuxlet_amd import build
    build.build()
    assert CLI.exists()
    return str(CLI)


def parse_dump(path):
    d = dict(sm=[], snps=[], g=[], cells=[])
    cur = None
    for line in open(path):
        t = line.rstrip("\n").split("\t")
        if t[0] == "SM": d["sm"].append(t[1])
        elif t[0] == "SNP":
            d["snps"].append((int(t[2]), int(t[3]), t[4], t[5]))
            d["g"].append([float.fromhex(x) for x in t[6:]])
        elif t[0] == "CELL":
            cur = dict(bc=t[2], cnt=tuple(map(int, t[3:6])), pairs=[])
            d["cells"].append(cur)
        elif t[0] == "PAIR":
            cur["pairs"].append((int(t[1]), [tuple(map(int, x.split(":"))) for x in t[3:]]))
    return d


def expected_from_scan(oracle, snps, events, gts, recs, field, gt_error):
    ev = oracle.Events([e[0] for e in events], np.array([e[1] for e in events], dtype=np.int32), [e[2] for e in events],
                       np.array([e[3] for e in events], dtype=np.uint8), np.array([e[4] for e in events], dtype=np.uint8),
                       np.array([e[5] for e in events], dtype=np.uint8))
    return oracle.store_from_events(ev)


def check_dump_against_scan(oracle, dump, snps, events, gts, recs, sm_cols, field="GT", gt_error=0.01):
    csr = expected_from_scan(oracle, snps, events, gts, recs, field, gt_error)
    assert [c["bc"] for c in dump["cells"]] == csr.barcodes
    assert len(dump["snps"]) == len(snps)
    for (rid, pos, ref, alt), i in zip(dump["snps"], snps):
        assert pos == recs[i]["pos"] and ref == recs[i]["ref"][0] and alt == recs[i]["alt"].split(",")[0][0]
    for c, cell in enumerate(dump["cells"]):
        assert cell["cnt"] == (csr.rd_totl[c], csr.rd_pass[c], csr.rd_uniq[c])
        p0, p1 = csr.cell_off[c], csr.cell_off[c + 1]
        assert [p[0] for p in cell["pairs"]] == list(csr.pair_snp[p0:p1])
        for (snp, rds), p in zip(cell["pairs"], range(p0, p1)):
            w = csr.words[csr.pair_off[p]:csr.pair_off[p + 1]]
            exp = [(int(x >> 24) & 0xFF, int(x >> 16) & 0xFF) for x in w if ((x >> 24) & 0xFF) != 2]
            assert rds == exp
    # genotype matrix through the oracle's a3 restatement
    for row, i, a in zip(dump["g"], snps, gts):
        if field == "GT":
            exp = oracle.geno_from_gt(np.array(a), gt_error)
        elif field == "PL":
            k = recs[i]["fmt"].index("PL") if "fmt" in recs[i] else 1            # (the synthetic VCFs are GT:PL:GP)
            pl = [[(np.iinfo(np.int32).min if x == "." else int(x)) for x in ((recs[i]["fields"][c].split(":") + ["."] * 8)[k].split(",") + [".", ".", "."])[:3]] for c in sm_cols]
            exp = oracle.geno_from_pl(np.array(pl))
        else:
            gp = [[(np.nan if x == "." else float(x)) for x in (recs[i]["fields"][c].split(":")[2].split(",") + [".", ".", "."])[:3]] for c in sm_cols]
            exp = oracle.geno_from_gp(np.array(gp, dtype=np.float32), gt_error)
        got = np.array(row, dtype=np.float64).reshape(-1, 3)
        assert np.array_equal(np.nan_to_num(got, nan=-1), np.nan_to_num(exp.astype(np.float64), nan=-1))


@pytest.mark.parametrize("field,fmt", [("GT", "sam"), ("PL", "bam"), ("GP", "sam")])
def test_scan_matches_restatement(cli, oracle, tmp_path, field, fmt):
    rng = np.random.default_rng(31 + len(field))
    recs = sv.make_vcf(rng, CONTIGS, 120, SAMPLES, tmp_path / "v.vcf.gz")
    reads = sv.make_reads(rng, CONTIGS, recs, 4000, [f"BC{i:02d}-1" for i in range(25)], tmp_path / "r.sam", tmp_path / "r.bam")
    out = tmp_path / "o"
    subprocess.run([cli, "--sam", str(tmp_path / f"r.{fmt}"), "--vcf", str(tmp_path / "v.vcf.gz"), "--field", field, "--out", str(out),
                    "--pileup-only"], check=True, stderr=subprocess.DEVNULL)
    dump = parse_dump(str(out) + ".pileup.txt")
    snps, events, gts, sm_cols = sv.scan(reads, recs, CONTIGS, SAMPLES)
    assert dump["sm"] == SAMPLES
    assert len(events) > 1000
    check_dump_against_scan(oracle, dump, snps, events, gts, recs, sm_cols, field)


def test_sam_and_bam_give_the_same_pileup(cli, tmp_path):
    """Every container the readers accept gives the same pileup: SAM text, gzip'd SAM, BAM in BGZF blocks (what samtools
    writes; inflated on several threads: block size forced down so that batches span many blocks), BAM as one ordinary
    gzip member, bgzip'd SAM and bgzip'd VCF."""
    import gzip
    rng = np.random.default_rng(5)
    recs = sv.make_vcf(rng, CONTIGS, 60, SAMPLES, tmp_path / "v.vcf")
    reads = sv.make_reads(rng, CONTIGS, recs, 1500, ["A-1", "C-1", "G-1"], tmp_path / "r.sam", tmp_path / "r.bam")
    sv.write_bam(CONTIGS, reads, tmp_path / "plain.bam", bgzf=False)
    sam_text = (tmp_path / "r.sam").read_bytes()
    (tmp_path / "r.sam.gz").write_bytes(gzip.compress(sam_text))
    (tmp_path / "rb.sam.gz").write_bytes(sv.bgzf_compress(sam_text, block=3000))          # hundreds of tiny BGZF blocks
    (tmp_path / "vb.vcf.gz").write_bytes(sv.bgzf_compress((tmp_path / "v.vcf").read_bytes(), block=777))
    raw_bam = gzip.decompress((tmp_path / "plain.bam").read_bytes())
    (tmp_path / "small.bam").write_bytes(sv.bgzf_compress(raw_bam, block=1021))          # records straddle block boundaries
    outs = {}
    runs = [("r.sam", "v.vcf", "1"), ("r.sam.gz", "v.vcf", "1"), ("rb.sam.gz", "v.vcf", "3"), ("r.bam", "v.vcf", "1"), ("r.bam", "vb.vcf.gz", "4"),
            ("plain.bam", "v.vcf", "2"), ("small.bam", "vb.vcf.gz", "7")]
    for i, (sam, vcf, threads) in enumerate(runs):
        env = dict(os.environ, DMX_THREADS=threads)
        subprocess.run([cli, "--sam", str(tmp_path / sam), "--vcf", str(tmp_path / vcf), "--field", "GT", "--out", str(tmp_path / f"o{i}"),
                        "--pileup-only"], check=True, stderr=subprocess.DEVNULL, env=env)
        outs[i] = (tmp_path / f"o{i}.pileup.txt").read_bytes()
    assert len(outs[0]) > 1000
    for i in outs:
        assert outs[i] == outs[0], runs[i]


@pytest.mark.parametrize("field", ["GT", "PL", "GP"])
def test_bcf_and_vcf_give_the_same_pileup(cli, tmp_path, field):
    """--vcf accepts BCF2 (bgzip'd binary VCF, read without htslib): the same records as VCF text and as BCF — integer vectors as
    int8, int16 or int32, records straddling BGZF blocks, explicit IDX= dictionary indices — give the same pileup, byte for byte,
    including the records the variant filter drops (missing calls, half-missing and haploid genotypes, multi-allelic sites)."""
    import gzip
    rng = np.random.default_rng(77)
    recs = sv.make_vcf(rng, CONTIGS, 90, SAMPLES, tmp_path / "v.vcf")
    reads = sv.make_reads(rng, CONTIGS, recs, 1500, ["A-1", "C-1", "G-1", "T-1"], tmp_path / "r.sam")
    text = (tmp_path / "v.vcf").read_text()
    # a few haploid and all-missing calls beside what make_vcf already draws
    lines = text.split("\n")
    k = 0
    for i, l in enumerate(lines):
        if l and not l.startswith("#"):
            f = l.split("\t")
            if k % 7 == 0: f[9] = "1:" + f[9].split(":", 1)[1] if ":" in f[9] else "1"
            if k % 11 == 0: f[10] = "."
            lines[i] = "\t".join(f)
            k += 1
    text = "\n".join(lines)
    (tmp_path / "v2.vcf").write_text(text)
    sv.vcf_text_to_bcf(text, tmp_path / "a.bcf", int_type=2)
    sv.vcf_text_to_bcf(text, tmp_path / "b.bcf", block=333, int_type=1 if field != "PL" else 3, idx_attrs=True)
    outs = []
    for v in ("v2.vcf", "a.bcf", "b.bcf"):
        o = tmp_path / ("o_" + v.split(".")[0])
        subprocess.run([cli, "--sam", str(tmp_path / "r.sam"), "--vcf", str(tmp_path / v), "--field", field, "--out", str(o), "--pileup-only"],
                       check=True, stderr=subprocess.DEVNULL)
        outs.append((str(o) + ".pileup.txt"))
    ref = open(outs[0], "rb").read()
    assert len(ref) > 1000
    for o in outs[1:]:
        assert open(o, "rb").read() == ref, o


def test_corrupt_bcf_records_never_crash_the_reader(cli, tmp_path):
    """Random byte damage inside a BCF (valid BGZF around it): the binary either reads through or stops with a fatal message
    (exit 1) — never a signal.  (An AddressSanitizer/UBSan build of the same source ran 300 such mutations clean.)"""
    import struct
    import zlib
    rng = np.random.default_rng(123)
    recs = sv.make_vcf(rng, CONTIGS, 40, SAMPLES, tmp_path / "v.vcf")
    sv.make_reads(rng, CONTIGS, recs, 300, ["A-1", "C-1"], tmp_path / "r.sam")
    sv.vcf_text_to_bcf((tmp_path / "v.vcf").read_text(), tmp_path / "a.bcf")
    bg = (tmp_path / "a.bcf").read_bytes()
    data, o = bytearray(), 0
    while o < len(bg):
        bsize = struct.unpack_from("<H", bg, o + 16)[0] + 1
        data += zlib.decompress(bg[o + 18:o + bsize - 8], -15)
        o += bsize
    hdr_len = 9 + struct.unpack_from("<I", data, 5)[0]
    codes = set()
    for it in range(60):
        d = bytearray(data)
        for _ in range(int(rng.integers(1, 5))):
            d[int(rng.integers(hdr_len, len(d)))] = int(rng.integers(0, 256))
        (tmp_path / "m.bcf").write_bytes(sv.bgzf_compress(bytes(d)))
        r = subprocess.run([cli, "--sam", str(tmp_path / "r.sam"), "--vcf", str(tmp_path / "m.bcf"), "--field", ["GT", "PL", "GP"][it % 3],
                            "--out", str(tmp_path / "o"), "--pileup-only"], capture_output=True)
        assert r.returncode in (0, 1), (it, r.returncode, r.stderr[-300:])
        codes.add(r.returncode)
    assert codes == {0, 1}                      # some damage is harmless, some is detected


def test_corrupt_bgzf_is_fatal(cli, tmp_path):
    """A flipped byte inside a BGZF block (CRC mismatch) and a truncated file stop the run with a message, as htslib would."""
    rng = np.random.default_rng(6)
    recs = sv.make_vcf(rng, CONTIGS, 30, SAMPLES, tmp_path / "v.vcf")
    sv.make_reads(rng, CONTIGS, recs, 800, ["A-1", "C-1"], tmp_path / "r.sam", tmp_path / "r.bam")
    good = bytearray((tmp_path / "r.bam").read_bytes())
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x5a
    (tmp_path / "flip.bam").write_bytes(bytes(bad))
    (tmp_path / "trunc.bam").write_bytes(bytes(good[: len(good) // 2]))
    for name in ("flip.bam", "trunc.bam"):
        r = subprocess.run([cli, "--sam", str(tmp_path / name), "--vcf", str(tmp_path / "v.vcf"), "--field", "GT", "--out", str(tmp_path / "x"),
                            "--pileup-only"], capture_output=True, text=True)
        assert r.returncode != 0 and ("BGZF" in r.stderr or "truncated" in r.stderr or "corrupt" in r.stderr), (name, r.stderr[-300:])


def test_options_filters_and_sample_selection(cli, oracle, tmp_path):
    """--sm goes through a std::set (sorted id order, bcf_filtered_reader.cpp:107-124); --min-BQ/--cap-BQ/--min-MQ/--min-TD/
    --excl-flag/--group-list change what reaches the store (cmd_cram_demuxlet.cpp:314-323, sam_filtered_reader.cpp:284-296)."""
    rng = np.random.default_rng(8)
    recs = sv.make_vcf(rng, CONTIGS, 100, SAMPLES, tmp_path / "v.vcf")
    bcs = [f"BC{i:02d}-1" for i in range(12)]
    reads = sv.make_reads(rng, CONTIGS, recs, 3000, bcs, tmp_path / "r.sam")
    (tmp_path / "grp.txt").write_text("\n".join(bcs[:5]) + "\n")
    out = tmp_path / "o"
    subprocess.run([cli, "--sam", str(tmp_path / "r.sam"), "--vcf", str(tmp_path / "v.vcf"), "--field", "GT", "--out", str(out), "--pileup-only",
                    "--sm", "smD", "--sm", "smA", "--sm", "smC", "--min-BQ", "20", "--cap-BQ", "30", "--min-MQ", "30", "--min-TD", "5",
                    "--excl-flag", "3860", "--group-list", str(tmp_path / "grp.txt"), "--geno-error", "0.05"], check=True, stderr=subprocess.DEVNULL)
    dump = parse_dump(str(out) + ".pileup.txt")
    assert dump["sm"] == ["smA", "smC", "smD"]
    snps, events, gts, sm_cols = sv.scan(reads, recs, CONTIGS, SAMPLES, sm_ids=["smD", "smA", "smC"], min_mq=30, excl_flag=3860, min_bq=20,
                                         cap_bq=30, min_td=5, group=set(bcs[:5]))
    check_dump_against_scan(oracle, dump, snps, events, gts, recs, sm_cols, "GT", 0.05)


def test_option_errors(cli, tmp_path):
    for args in (["--no-such-option"], ["--min-BQ", "abc"], ["--sam", "a", "--sam", "b"], ["stray"], ["--out", "x"]):
        r = subprocess.run([cli] + args, capture_output=True, text=True)
        assert r.returncode != 0 and "FATAL ERROR" in r.stderr, args
    r = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert r.returncode == 1 and "--write-pair" in r.stderr and "--doublet-prior" in r.stderr      # params.cpp:457-463: help exits 1


def test_tutorial_vcf_plumbing(cli, oracle, tmp_path):
    """BASELINE config 1 plumbing: the reference tutorial's VCF (first 4000 records, a data fixture) + a synthetic SAM laid over
    it -> the scan keeps exactly the biallelic records with call rate >= 0.5 and MAC >= 1 among jurkat/293T_RTG."""
    import gzip
    vcf = ROOT / "tests" / "golden" / "tutorial_jurkat_293T_first4000.vcf.gz"
    recs, contigs = [], []
    for line in gzip.open(vcf, "rt"):
        if line.startswith("##contig=<ID="):
            name = line[13:].split(",")[0].rstrip(">\n")
            ln = int(line.split("length=")[1].split(">")[0].split(",")[0]) if "length=" in line else 250000000
            contigs.append((name, ln))
        elif not line.startswith("#"):
            t = line.rstrip("\n").split("\t")
            recs.append(dict(chrom=t[0], pos=int(t[1]) - 1, ref=t[3], alt=t[4], fields=t[9:]))
    rng = np.random.default_rng(2)
    used = [c for c in contigs if c[0] in {r["chrom"] for r in recs}]
    reads = sv.make_reads(rng, used, recs, 3000, [f"CELL{i:03d}-1" for i in range(40)], tmp_path / "r.sam")
    # make_reads' tids index `used`; write the header with the same list
    subprocess.run([cli, "--sam", str(tmp_path / "r.sam"), "--vcf", str(vcf), "--field", "GT", "--out", str(tmp_path / "o"), "--pileup-only"],
                   check=True, stderr=subprocess.DEVNULL)
    dump = parse_dump(str(tmp_path / "o.pileup.txt"))
    assert dump["sm"] == ["jurkat", "293T_RTG"]
    snps, events, gts, sm_cols = sv.scan(reads, recs, used, ["jurkat", "293T_RTG"])
    assert len(dump["snps"]) == len(snps) and len(snps) > 500
    assert sum(len(c["pairs"]) for c in dump["cells"]) > 500
    check_dump_against_scan(oracle, dump, snps, events, gts, recs, sm_cols, "GT")
    # the same file as BCF2 (a real-world header: INFO / FILTER / FORMAT dictionaries, many contigs): the same pileup
    sv.vcf_text_to_bcf(gzip.open(vcf, "rt").read(), tmp_path / "t.bcf")
    subprocess.run([cli, "--sam", str(tmp_path / "r.sam"), "--vcf", str(tmp_path / "t.bcf"), "--field", "GT", "--out", str(tmp_path / "ob"), "--pileup-only"],
                   check=True, stderr=subprocess.DEVNULL)
    assert (tmp_path / "ob.pileup.txt").read_bytes() == (tmp_path / "o.pileup.txt").read_bytes()


HELP_HEAD = """
Detailed instructions of parameters are available. Ones with "[]" are in effect:

Available Options


Options for input SAM/BAM/CRAM
  --sam           [STR: ]             : Input SAM/BAM/CRAM file. Must be sorted by coordinates and indexed
  --tag-group     [STR: CB]           : Tag representing readgroup or cell barcodes, in the case to partition the BAM file into multiple groups. For 10x genomics, use CB
  --tag-UMI       [STR: UB]           : Tag representing UMIs. For 10x genomiucs, use UB

Options for input VCF/BCF
  --vcf           [STR: ]             : Input VCF/BCF file, containing the individual genotypes (GT), posterior probability (GP), or genotype likelihood (PL)
  --field         [STR: GP]           : FORMAT field to extract the genotype, likelihood, or posterior from
  --geno-error    [FLT: 0.01]         : Genotype error rate (must be used with --field GT)
  --min-mac       [INT: 1]            : Minimum minor allele frequency
  --min-callrate  [FLT: 0.50]         : Minimum call rate
  --sm            [V_STR: ]           : List of sample IDs to compare to (default: use all)
  --sm-list       [STR: ]             : File containing the list of sample IDs to compare

Output Options
  --out           [STR: ]             : Output file prefix
  --alpha         [V_FLT: ]           : Grid of alpha to search for (default is 0, 0.5)
  --write-pair    [FLG: OFF]          : Writing the (HUGE) pair file
  --doublet-prior [FLT: 0.50]         : Prior of doublet
  --sam-verbose   [INT: 1000000]      : Verbose message frequency for SAM/BAM/CRAM
  --vcf-verbose   [INT: 10000]        : Verbose message frequency for VCF/BCF

Read filtering Options
  --cap-BQ        [INT: 40]           : Maximum base quality (higher BQ will be capped)
  --min-BQ        [INT: 13]           : Minimum base quality to consider (lower BQ will be skipped)
  --min-MQ        [INT: 20]           : Minimum mapping quality to consider (lower MQ will be ignored)
  --min-TD        [INT: 0]            : Minimum distance to the tail (lower will be ignored)
  --excl-flag     [INT: 3844]         : SAM/BAM FLAGs to be excluded

Cell/droplet filtering options
  --group-list    [STR: ]             : List of tag readgroup/cell barcode to consider in this run. All other barcodes will be ignored. This is useful for parallelized run
  --min-total     [INT: 0]            : Minimum number of total reads for a droplet/cell to be considered
  --min-uniq      [INT: 0]            : Minimum number of unique reads (determined by UMI/SNP pair) for a droplet/cell to be considered
  --min-snp       [INT: 0]            : Minimum number of SNPs with coverage for a droplet/cell to be considered
"""

STATUS_HEAD = """
Available Options

The following parameters are available. Ones with "[]" are in effect:
   Options for input SAM/BAM/CRAM : --sam [r.sam], --tag-group [CB],
                                    --tag-UMI [UB]
        Options for input VCF/BCF : --vcf [v.vcf], --field [GT],
                                    --geno-error [1.0e-03], --min-mac [1],
                                    --min-callrate [0.50], --sm [smA, smC],
                                    --sm-list
                   Output Options : --out [o], --alpha [0.00, 0.25, 0.50],
                                    --write-pair [ON], --doublet-prior [0.50],
                                    --sam-verbose [1000000],
                                    --vcf-verbose [10000]
           Read filtering Options : --cap-BQ [40], --min-BQ [13],
                                    --min-MQ [20], --min-TD,
                                    --excl-flag [3844]
   Cell/droplet filtering options : --group-list, --min-total, --min-uniq,
                                    --min-snp [5]
"""


def test_help_and_status_text_are_the_reference_layout(cli, tmp_path):
    """f4: `--help` (paramList::HelpMessage + longParams::HelpMessage, params.cpp:306-405,:527-550) and the status echo every run
    prints (paramList::Status + longParams::Status, :188-303,:552-574), compared with literal text laid out by hand from those
    functions: "  --%-*s%-*s : help" with the name column = longest option name (13, doublet-prior) and a 20-column state, groups
    right-aligned to the longest group title + 2, items wrapped at 78 columns with a continuation indent of group_len + 5,
    doubles "%.2f" (or "%.1e" below 0.01), zero / empty values echoed without brackets.  This build's own options form one more
    group after the reference's; it is shorter than either column and does not move them."""
    r = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout == ""
    assert r.stderr.startswith(HELP_HEAD + "\nMI355X build\n"), r.stderr[:400]
    assert r.stderr.endswith("\n\n\nNOTES:\nWhen --help was included in the argument. The program prints the help message but do not actually run\n")
    (tmp_path / "r.sam").write_text("@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:1\tLN:1000\n")
    (tmp_path / "v.vcf").write_text("##fileformat=VCFv4.2\n##contig=<ID=1,length=1000>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsmA\tsmC\n")
    r = subprocess.run([cli, "--sam", "r.sam", "--vcf", "v.vcf", "--field", "GT", "--geno-error", "0.001", "--sm", "smA", "--sm", "smC", "--out", "o",
                        "--alpha", "0", "--alpha", "0.25", "--alpha", "0.5", "--write-pair", "--min-snp", "5", "--pileup-only"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.stderr.startswith(STATUS_HEAD + "                     MI355X build : --gpu, --gpus [1], --pileup-only [ON],\n"
                                             "                                    --no-arbiter, --strict, --fast\n"), r.stderr[:1500]
    assert "\n\nRun with --help for more detailed help messages of each argument.\n\n" in r.stderr
    # parse errors are reported AFTER the status echo, as paramList::Status does (:562-567)
    r = subprocess.run([cli, "--sam", "r.sam", "--bogus"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0 and r.stderr.index("Run with --help") < r.stderr.index("Command line parameter --bogus (#3) not recognized")


REF_PARAMS = ROOT / "oracle" / "_ref" / "ref_params_driver"


def _strip_own_group(text):
    """Our binary adds ONE option group ("MI355X build") behind the reference's: drop it from the status echo / the help text."""
    out, skip = [], False
    for ln in text.split("\n"):
        if ln.strip().startswith("MI355X build :") or ln == "MI355X build":
            skip = True                                   # status: the group's first line; help: its title line
            if out and out[-1] == "" and ln == "MI355X build":
                out.pop()                                 # help: the blank line in front of the title
            continue
        if skip:
            if ln.startswith("  --") or (ln.startswith(" " * 30) and ln.strip().startswith("--")):
                continue                                  # help rows / status continuation rows of that group
            skip = False
        out.append(ln)
    return "\n".join(out)


@pytest.mark.parametrize("args", [
    ["--help"],
    ["--sam", "r.sam", "--vcf", "v.vcf", "--field", "GT", "--out", "o", "--pileup-only"],
    ["--sam", "r.sam", "--vcf", "v.vcf", "--field", "GT", "--geno-error", "0.001", "--sm", "smA", "--sm", "smC", "--out", "o", "--alpha", "0", "--alpha",
     "0.25", "--alpha", "0.5", "--write-pair", "--min-snp", "5", "--pileup-only"],
    ["--sam", "r.sam", "--vcf", "v.vcf", "--out", "some/long/prefix/for/the/output/files", "--tag-group", "XC", "--tag-UMI", "XM", "--min-mac", "3",
     "--min-callrate", "0.95", "--doublet-prior", "0.0625", "--cap-BQ", "30", "--min-BQ", "20", "--min-MQ", "30", "--min-TD", "5", "--excl-flag", "1796",
     "--min-total", "100", "--min-uniq", "50", "--sam-verbose", "5000", "--vcf-verbose", "7", "--geno-error", "0.00001", "--pileup-only"],
    ["--sam", "r.sam", "--bogus"], ["--sam", "a", "--sam", "b"], ["--min-mac", "x1"], ["--min-mac", "x1", "--write-pair", "3", "--write-pair"],
    ["--alpha", "zero", "--geno-error", "1e-3", "--geno-error", "2", "-x", "--out"], ["--sm"], ["--min-snp"]])
def test_help_status_and_errors_equal_the_reference_parser(cli, tmp_path, args):
    """f4 pinned by the reference itself: params.cpp + Error.cpp compiled alone (oracle/Makefile: ref_params_driver) and driven by the
    reference's own option table (cmd_cram_demuxlet.cpp:9-76) print the --help text, the status echo and the option errors; the
    `demuxlet` binary must print the same characters — apart from its one extra option group — and exit the same way."""
    if not REF_PARAMS.exists():
        pytest.skip("oracle/_ref/ref_params_driver not built (needs /root/reference)")
    (tmp_path / "r.sam").write_text("@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:1\tLN:1000\n")
    (tmp_path / "v.vcf").write_text("##fileformat=VCFv4.2\n##contig=<ID=1,length=1000>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsmA\tsmC\n")
    ref_args = [a for a in args if a != "--pileup-only"]
    want = subprocess.run([str(REF_PARAMS)] + ref_args, capture_output=True, text=True, cwd=tmp_path)
    got = subprocess.run([cli] + args, capture_output=True, text=True, cwd=tmp_path)
    w = want.stderr
    g = _strip_own_group(got.stderr)
    if "--help" in args:
        assert got.returncode == 1 and got.stdout == want.stdout == ""
        assert g == w
        return
    # the status echo is the first thing either program prints; ours goes on with the scan's own messages
    head = w[:w.index("Run with --help for more detailed help messages of each argument.\n\n") + len("Run with --help for more detailed help messages of each argument.\n\n")]
    assert g.startswith(head), (g[:600], head[:600])
    if want.returncode != 0:                                # option errors: same wording, same place (after the echo), failure status
        assert got.returncode != 0
        import re
        tail = w[len(head):]
        # the reference's messages carry its __FILE__ ("[E:<the path it was compiled from>/params.cpp:564 Status] "): the directory part is
        # the build's, not the program's; its uncaught exception makes the C++ runtime add two lines of its own
        tail = re.sub(r"\[E:[^\]:]*/params\.cpp:", "[E:params.cpp:", tail)
        tail = tail.split("terminate called after throwing")[0]
        assert tail.strip() and g[len(head):].startswith(tail.rstrip("\n")), (g[len(head):][:400], tail[:400])


@pytest.mark.parametrize("fmt,field,extra", [("bam", "GT", []), ("sam", "PL", ["--min-TD", "3", "--min-BQ", "20", "--cap-BQ", "35"]),
                                             ("bam", "GP", ["--group-list", "groups.txt"]), ("sam", "GT", ["--tag-group", "XX", "--tag-UMI", "YY"])])
def test_windowed_scan_equals_the_read_by_read_scan(cli, tmp_path, fmt, field, extra):
    """f1 on several host threads: windows of reads parsed, overlapped with the SNPs and stored in parallel (three pipeline stages,
    cell-sharded store batches) must leave the SAME pileup, counters and messages as the read-by-read scan on one thread — whatever the
    window size, also with a barcode list, missing tags (every read in one group / one UMI) and a VCF read ahead on its own thread."""
    rng = np.random.default_rng(4242 + len(field) + len(extra))
    recs = sv.make_vcf(rng, CONTIGS, 200, SAMPLES, tmp_path / "v.vcf.gz", with_noise=(field != "GP"))
    bcs = [f"BC{i:02d}-1" for i in range(30)]
    sv.make_reads(rng, CONTIGS, recs, 5000, bcs, tmp_path / "r.sam", tmp_path / "r.bam")
    (tmp_path / "groups.txt").write_text("\n".join(bcs[::3]) + "\n")
    outs = []
    for i, env_extra in enumerate(({"DMX_THREADS": "1"}, {"DMX_THREADS": "4", "DMX_SCAN_WINDOW": "37"}, {"DMX_THREADS": "3", "DMX_SCAN_WINDOW": "1000"},
                                   {"DMX_THREADS": "6"}, {"DMX_THREADS": "5", "DMX_SCAN_SEQUENTIAL": "1"})):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([cli, "--sam", f"r.{fmt}", "--vcf", "v.vcf.gz", "--field", field, "--out", f"o{i}", "--pileup-only"] + extra,
                           capture_output=True, text=True, cwd=tmp_path, env=env)
        assert r.returncode == 0, r.stderr[-400:]
        totals = [ln.split("] - ", 1)[1] for ln in r.stderr.splitlines() if "Total number" in ln or "Finished reading" in ln]
        outs.append(((tmp_path / f"o{i}.pileup.txt").read_bytes(), totals))
    assert len(outs[0][1]) >= 10 and b"PAIR" in outs[0][0]
    for got in outs[1:]:
        assert got[0] == outs[0][0]
        assert got[1] == outs[0][1]


@pytest.mark.parametrize("fmt", ["bam", "sam"])
def test_windowed_scan_with_a_vcf_that_has_no_contig_lines(cli, tmp_path, fmt):
    """a VCF without ##contig lines registers its contigs record by record — in the windowed scan on the feed thread,
    which parses ahead of the reads.  The scanning thread must learn them only through the records it consumes (a read on a contig the
    VCF has not reached yet is skipped, cmd_cram_demuxlet.cpp:198-200): same pileup, counters and messages as read by read on one
    thread, run after run."""
    rng = np.random.default_rng(977)
    contigs = [("1", 20000), ("2", 15000), ("X", 12000), ("7", 9000)]
    recs = sv.make_vcf(rng, contigs, 120, SAMPLES, tmp_path / "v.vcf.gz", with_noise=True, contig_lines=False)
    bcs = [f"BC{i:02d}-1" for i in range(24)]
    sv.make_reads(rng, contigs, recs, 6000, bcs, tmp_path / "r.sam", tmp_path / "r.bam")
    outs = []
    runs = [{"DMX_THREADS": "1"}] + [{"DMX_THREADS": "4", "DMX_SCAN_WINDOW": "53"}] * 4 + [{"DMX_THREADS": "6"}] * 3 + [{"DMX_THREADS": "3", "DMX_SCAN_SEQUENTIAL": "1"}]
    for i, env_extra in enumerate(runs):
        r = subprocess.run([cli, "--sam", f"r.{fmt}", "--vcf", "v.vcf.gz", "--field", "GT", "--out", f"o{i}", "--pileup-only"],
                           capture_output=True, text=True, cwd=tmp_path, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stderr[-400:]
        totals = [ln.split("] - ", 1)[1] for ln in r.stderr.splitlines() if "Total number" in ln or "Finished reading" in ln]
        outs.append(((tmp_path / f"o{i}.pileup.txt").read_bytes(), totals))
    assert len(outs[0][1]) >= 10 and b"PAIR" in outs[0][0]
    for got in outs[1:]:
        assert got[0] == outs[0][0] and got[1] == outs[0][1]


def test_tutorial_vcf_at_full_size_plumbing(cli, oracle, tmp_path):
    """BASELINE config 1's VCF at its size on the host side: all 54 424 records of the tutorial file (a data fixture), reads over all of
    its 24 contigs.  --field GT: the pileup and the genotype matrix are the restatement's, incl. the 25 259 records whose second sample
    is missing (Hardy-Weinberg fallback, bcf_filtered_reader.cpp:381-388).  --field PL: 16 584 records carry no PL and the reference is
    fatal at the first one the scan reaches (cmd_cram_demuxlet.cpp:211-212) — same message here; on the 37 840 records that do carry PL
    the matrix is the 10-iteration EM of :244-320 on the file's own values."""
    import gzip
    vcf = ROOT / "tests" / "golden" / "tutorial_jurkat_293T_exons_only.vcf.gz"
    recs, contigs = [], []
    for line in gzip.open(vcf, "rt"):
        if line.startswith("##contig=<ID="):
            name = line[13:].split(",")[0].rstrip(">\n")
            contigs.append((name, int(line.split("length=")[1].split(">")[0].split(",")[0]) if "length=" in line else 250000000))
        elif not line.startswith("#"):
            t = line.rstrip("\n").split("\t")
            recs.append(dict(chrom=t[0], pos=int(t[1]) - 1, ref=t[3], alt=t[4], fields=t[9:], fmt=t[8].split(":")))
    assert len(recs) == 54424
    used = [c for c in contigs if c[0] in {r["chrom"] for r in recs}]
    rng = np.random.default_rng(2025)
    reads = sv.make_reads(rng, used, recs, 40000, [f"CELL{i:03d}-1" for i in range(100)], tmp_path / "r.sam", tmp_path / "r.bam")
    samples = ["jurkat", "293T_RTG"]
    subprocess.run([cli, "--sam", str(tmp_path / "r.bam"), "--vcf", str(vcf), "--field", "GT", "--out", str(tmp_path / "o"), "--pileup-only"],
                   check=True, stderr=subprocess.DEVNULL)
    dump = parse_dump(str(tmp_path / "o.pileup.txt"))
    snps, events, gts, sm_cols = sv.scan(reads, recs, used, samples)
    assert len(snps) > 50000 and sum(1 for a in gts if any(x < 0 for x in a[1])) == 25259
    check_dump_against_scan(oracle, dump, snps, events, gts, recs, sm_cols, "GT")
    first_without = next(r for r in recs if "PL" not in r["fmt"])
    r = subprocess.run([cli, "--sam", str(tmp_path / "r.bam"), "--vcf", str(vcf), "--field", "PL", "--out", str(tmp_path / "x"), "--pileup-only"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and f"Cannot parse posterior probability at {first_without['chrom']}:{first_without['pos'] + 1}" in r.stderr
    with gzip.open(tmp_path / "pl.vcf.gz", "wt") as f:
        f.writelines(l for l in gzip.open(vcf, "rt") if l.startswith("#") or "PL" in l.split("\t")[8].split(":"))
    recs_pl = [r for r in recs if "PL" in r["fmt"]]
    assert len(recs_pl) == 37840
    subprocess.run([cli, "--sam", str(tmp_path / "r.bam"), "--vcf", str(tmp_path / "pl.vcf.gz"), "--field", "PL", "--out", str(tmp_path / "p"), "--pileup-only"],
                   check=True, stderr=subprocess.DEVNULL)
    snps2, events2, gts2, sm2 = sv.scan(reads, recs_pl, used, samples)
    check_dump_against_scan(oracle, parse_dump(str(tmp_path / "p.pileup.txt")), snps2, events2, gts2, recs_pl, sm2, "PL")
