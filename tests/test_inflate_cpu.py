"""CPU: the scanner's own DEFLATE decoder and CRC-32 (demuxlet_amd/csrc/dmx_inflate.hpp, used for BGZF members of BAM / bgzipped
VCF input, f2) against zlib — under AddressSanitizer + UBSan, since the decoder writes match copies in whole words and reads its
input ahead — and the `demuxlet` binary's scan with and without it."""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import sam_vcf_synth as sv

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("flags", [["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"], ["-O2"]])
def test_inflate_and_crc_against_zlib(tmp_path, flags):
    exe = tmp_path / "inflate_check"
    subprocess.run(["g++", "-std=c++17", *flags, "-I", str(ROOT / "demuxlet_amd" / "csrc"), str(ROOT / "tests" / "inflate_check.cpp"), "-o", str(exe), "-lz"],
                   check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "failures 0" in r.stdout
    n_ok = int(r.stdout.split("streams ok ")[1].split(",")[0])
    assert n_ok == 6 * 13 * 5 * 4 + 3            # + the three hand-made incomplete-distance-code streams (zlib takes L = 1 only)


def test_scan_is_the_same_through_zlib_and_through_our_decoder(tmp_path):
    """BAM + bgzipped VCF through the binary (--pileup-only dump of the store) with DMX_ZLIB_INFLATE=1 (zlib inflates every BGZF
    member) and without (dmx_inflate.hpp first, zlib only as the fallback): same bytes."""
    from demuxlet_amd import build
    build.build()
    rng = np.random.default_rng(5)
    contigs = [("1", 60000), ("2", 30000)]
    samples = ["a", "b", "c"]
    recs = sv.make_vcf(rng, contigs, 400, samples, tmp_path / "v.vcf.gz", with_noise=True)
    sv.make_reads(rng, contigs, recs, 30000, [f"BC{i:03d}-1" for i in range(60)], tmp_path / "r.sam", tmp_path / "r.bam")
    outs = []
    for name, env in (("zlib", {"DMX_ZLIB_INFLATE": "1"}), ("ours", {}), ("ours1", {"DMX_THREADS": "1"})):
        out = tmp_path / name
        subprocess.run([str(ROOT / "demuxlet_amd" / "demuxlet"), "--sam", str(tmp_path / "r.bam"), "--vcf", str(tmp_path / "v.vcf.gz"), "--field", "GT",
                        "--out", str(out), "--pileup-only"], check=True, env={**os.environ, **env}, stderr=subprocess.DEVNULL)
        outs.append(Path(f"{out}.pileup.txt").read_bytes())
    assert len(outs[0]) > 1000 and outs[0] == outs[1] == outs[2]


def test_corrupt_bgzf_member_is_reported(tmp_path):
    """A BAM whose BGZF member was damaged (one byte of deflate data flipped): our decoder refuses or the CRC disagrees, zlib is
    asked, and the binary stops with the BGZF error — it never scans garbage."""
    from demuxlet_amd import build
    build.build()
    rng = np.random.default_rng(6)
    contigs = [("1", 20000)]
    recs = sv.make_vcf(rng, contigs, 50, ["a", "b"], tmp_path / "v.vcf.gz", with_noise=False)
    sv.make_reads(rng, contigs, recs, 3000, ["BC-1", "BD-1"], tmp_path / "r.sam", tmp_path / "r.bam")
    raw = bytearray((tmp_path / "r.bam").read_bytes())
    raw[len(raw) // 2] ^= 0x10
    (tmp_path / "bad.bam").write_bytes(bytes(raw))
    r = subprocess.run([str(ROOT / "demuxlet_amd" / "demuxlet"), "--sam", str(tmp_path / "bad.bam"), "--vcf", str(tmp_path / "v.vcf.gz"), "--field", "GT",
                        "--out", str(tmp_path / "o"), "--pileup-only"], capture_output=True, text=True)
    assert r.returncode != 0
    assert "BGZF" in r.stderr
