"""GPU (-m gpu): the `demuxlet` binary end to end — SAM/BAM + VCF in, the four files out — against the oracle run on the
events of the independent scan restatement (tests/sam_vcf_synth.py)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import sam_vcf_synth as sv

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
CLI = ROOT / "demuxlet_amd" / "demuxlet"
CONTIGS = [("1", 30000), ("2", 20000), ("X", 15000)]
SAMPLES = ["smC", "smA", "smB", "smD"]


def compare_files(got_path, want_path, best=False):
    got = Path(got_path).read_text().splitlines()
    want = Path(want_path).read_text().splitlines()
    assert len(got) == len(want) and got[0] == want[0]
    for a, b in zip(got[1:], want[1:]):
        fa, fb = a.split("\t"), b.split("\t")
        assert len(fa) == len(fb)
        for x, y in zip(fa, fb):
            try:
                fx, fy = float(x), float(y)
                assert abs(fx - fy) <= 1e-3 * max(1e-3, abs(fy)) + 1.01e-4, (a, b)
            except ValueError:
                assert x == y, (a, b)


@pytest.mark.parametrize("field,fmt,extra", [("GT", "bam", ["--write-pair"]), ("GP", "sam", ["--alpha", "0", "--alpha", "0.25", "--alpha", "0.5"]),
                                              ("PL", "sam", ["--min-snp", "5", "--doublet-prior", "0.3"]),
                                              ("PL", "bam", ["--write-pair", "--gpus", "3"]),        # three engines (one device here)
                                              ("PL", "sam", ["--write-pair", "--fast", "--gpus", "2"]),    # DMX_MODE_FAST (opt-in)
                                              ("GT", "sam", ["--fast"]), ("GP", "bam", ["--fast"]),
                                              ("GP", "sam", ["--alpha", "0", "--alpha", "0.25", "--alpha", "0.5", "--fast"]),   # k_doublet_anf
                                              ("PL", "bam", ["--alpha", "0", "--alpha", "0.3", "--fast", "--write-pair"]),
                                              ("GT", "sam", ["--strict"])])                                 # the default (DMX_MODE_STRICT), spelled out
def test_cli_end_to_end(oracle, tmp_path, field, fmt, extra):
    from demuxlet_amd import build
    build.build()
    rng = np.random.default_rng(77 + len(field) + len(extra))
    recs = sv.make_vcf(rng, CONTIGS, 150, SAMPLES, tmp_path / "v.vcf.gz", with_noise=(field != "GP"))
    reads = sv.make_reads(rng, CONTIGS, recs, 6000, [f"BC{i:02d}-1" for i in range(20)], tmp_path / "r.sam", tmp_path / "r.bam")
    out = tmp_path / "o"
    subprocess.run([str(CLI), "--sam", str(tmp_path / f"r.{fmt}"), "--vcf", str(tmp_path / "v.vcf.gz"), "--field", field, "--out", str(out)] + extra,
                   check=True, stderr=subprocess.DEVNULL)
    # expectation: scan restatement -> oracle
    snps, events, gts, sm_cols = sv.scan(reads, recs, CONTIGS, SAMPLES)
    if field == "GT":
        g = np.stack([oracle.geno_from_gt(np.array(a), 0.01) for a in gts])
    elif field == "PL":
        g = np.stack([oracle.geno_from_pl(np.array([[(np.iinfo(np.int32).min if x == "." else int(x)) for x in (recs[i]["fields"][c].split(":")[1].split(",") + ["."] * 3)[:3]]
                                                    for c in sm_cols])) for i in snps])
    else:
        g = np.stack([oracle.geno_from_gp(np.array([[(np.nan if x == "." else float(x)) for x in (recs[i]["fields"][c].split(":")[2].split(",") + ["."] * 3)[:3]]
                                                    for c in sm_cols], dtype=np.float32), 0.01) for i in snps])
    ev = oracle.Events([e[0] for e in events], np.array([e[1] for e in events], dtype=np.int32), [e[2] for e in events],
                       np.array([e[3] for e in events], dtype=np.uint8), np.array([e[4] for e in events], dtype=np.uint8),
                       np.array([e[5] for e in events], dtype=np.uint8))
    alphas = tuple(float(extra[i + 1]) for i, x in enumerate(extra) if x == "--alpha") or (0.0, 0.5)
    params = oracle.Params(alphas=alphas, write_pair="--write-pair" in extra,
                           min_snp=int(extra[extra.index("--min-snp") + 1]) if "--min-snp" in extra else 0,
                           doublet_prior=float(extra[extra.index("--doublet-prior") + 1]) if "--doublet-prior" in extra else 0.5)
    pb = oracle.Problem(SAMPLES, np.nan_to_num(g.astype(np.float32)), ev, params)
    oracle.run_problem(pb, str(tmp_path / "orc"))
    for suf in ["single", "sing2", "best"] + (["pair"] if "--write-pair" in extra else []):
        compare_files(f"{out}.{suf}", tmp_path / f"orc.{suf}")
    got_best = [l.split("\t")[5] for l in Path(f"{out}.best").read_text().splitlines()]
    want_best = [l.split("\t")[5] for l in (tmp_path / "orc.best").read_text().splitlines()]
    assert got_best == want_best


def test_nan_likelihoods_do_not_crash(tmp_path):
    """A GP record with a missing sample turns the whole SNP into NaN (bcf_filtered_reader.cpp:431-448) and the reference then
    indexes its grid with -1; the product must neither fault on the GPU nor on the host, and must still write all files."""
    from demuxlet_amd import build
    build.build()
    rng = np.random.default_rng(3)
    recs = sv.make_vcf(rng, CONTIGS, 100, SAMPLES, tmp_path / "v.vcf.gz", with_noise=True)
    sv.make_reads(rng, CONTIGS, recs, 3000, [f"BC{i:02d}-1" for i in range(10)], tmp_path / "r.sam")
    r = subprocess.run([str(CLI), "--sam", str(tmp_path / "r.sam"), "--vcf", str(tmp_path / "v.vcf.gz"), "--field", "GP", "--out", str(tmp_path / "o"),
                        "--write-pair"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    for suf in ("single", "sing2", "best", "pair"):
        assert (tmp_path / f"o.{suf}").stat().st_size > 100


def test_cfg1_tutorial_vcf_through_the_binary_on_the_gpu(oracle, tmp_path):
    """BASELINE config 1 on the HIP path: the reference tutorial's VCF (tutorial/README.MD; first 4 000 records, a data fixture —
    the tutorial BAM is not in the reference repository) + a synthetic 500-barcode SAM laid over it, `--field GT --alpha 0
    --alpha 0.5` through the `demuxlet` binary on the GPU, against the oracle's four files for the same scan."""
    import gzip
    from demuxlet_amd import build
    build.build()
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
    rng = np.random.default_rng(20)
    used = [c for c in contigs if c[0] in {r["chrom"] for r in recs}]
    samples = ["jurkat", "293T_RTG"]
    reads = sv.make_reads(rng, used, recs, 40000, [f"CELL{i:03d}-1" for i in range(500)], tmp_path / "r.sam", tmp_path / "r.bam")
    out = tmp_path / "o"
    subprocess.run([str(CLI), "--sam", str(tmp_path / "r.bam"), "--vcf", str(vcf), "--field", "GT", "--alpha", "0", "--alpha", "0.5",
                    "--out", str(out)], check=True, stderr=subprocess.DEVNULL)
    snps, events, gts, sm_cols = sv.scan(reads, recs, used, samples)
    assert len(snps) > 500
    g = np.stack([oracle.geno_from_gt(np.array(a), 0.01) for a in gts])
    ev = oracle.Events([e[0] for e in events], np.array([e[1] for e in events], dtype=np.int32), [e[2] for e in events],
                       np.array([e[3] for e in events], dtype=np.uint8), np.array([e[4] for e in events], dtype=np.uint8),
                       np.array([e[5] for e in events], dtype=np.uint8))
    oracle.run_problem(oracle.Problem(samples, g.astype(np.float32), ev, oracle.Params()), str(tmp_path / "orc"))
    for suf in ("single", "sing2", "best"):
        compare_files(f"{out}.{suf}", tmp_path / f"orc.{suf}")
    got = [l.split("\t") for l in Path(f"{out}.best").read_text().splitlines()]
    want = [l.split("\t") for l in (tmp_path / "orc.best").read_text().splitlines()]
    assert len(got) > 300 and [r[5] for r in got] == [r[5] for r in want]         # every BEST call
    assert [r[:5] for r in got] == [r[:5] for r in want]                            # barcodes and read/SNP counters
    calls = {r[5][:3] for r in got[1:]}
    assert "SNG" in calls                                                          # the job is not degenerate
    # the same genotypes as BCF2: the same three files, byte for byte
    sv.vcf_text_to_bcf(gzip.open(vcf, "rt").read(), tmp_path / "t.bcf")
    outb = tmp_path / "ob"
    subprocess.run([str(CLI), "--sam", str(tmp_path / "r.bam"), "--vcf", str(tmp_path / "t.bcf"), "--field", "GT", "--alpha", "0", "--alpha", "0.5",
                    "--out", str(outb)], check=True, stderr=subprocess.DEVNULL)
    for suf in ("single", "sing2", "best"):
        assert Path(f"{outb}.{suf}").read_bytes() == Path(f"{out}.{suf}").read_bytes(), suf
