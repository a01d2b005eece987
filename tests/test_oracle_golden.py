"""CPU: the oracle (oracle/dmx_oracle.c) against the reference's own outputs (tests/golden/*.npz).
Bit-exact raw binary64 arrays, byte-identical .single/.sing2/.best/.pair."""
import numpy as np
import pytest

from golden_util import CASES, GOLDEN, Golden


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_outputs(oracle, name, tmp_path):
    gd = Golden(name)
    pb = gd.problem(oracle)
    csr, raw = oracle.run_problem(pb, str(tmp_path / "orc"))
    # a1: ids, counters, N.SNP
    assert csr.barcodes == gd.ref_barcodes
    cnt = gd.z["ref_counters"]
    assert np.array_equal(csr.rd_totl, cnt[:, 0]) and np.array_equal(csr.rd_pass, cnt[:, 1]) and np.array_equal(csr.rd_uniq, cnt[:, 2])
    assert np.array_equal(np.diff(csr.cell_off), cnt[:, 3])
    # a5
    assert np.array_equal(raw.llks, gd.z["ref_llks"])
    assert np.array_equal(raw.llk0s, gd.z["ref_llk0s"])
    # a8/a9
    assert np.array_equal(raw.processed, gd.z["ref_processed"])
    assert np.array_equal(raw.llksAB, gd.z["ref_llksAB"])
    assert np.array_equal(raw.llks00, gd.z["ref_llks00"])
    # a6, a10..a14: the four files
    for suf, ref_bytes in gd.files.items():
        got = (tmp_path / f"orc.{suf}").read_bytes()
        assert got == ref_bytes, f".{suf} differs"
    assert set(gd.files) == ({"single", "sing2", "best", "pair"} if gd.write_pair else {"single", "sing2", "best"})


def test_kat_best_rows_are_the_survey_ones():
    """SURVEY.md §4 known-answer vector, as literal text."""
    gd = Golden("kat_micro")
    best = gd.files["best"].decode().splitlines()
    assert best[1] == "AAA\t4\t4\t4\t3\tSNG-S0\tS0\t-0.0249\tS1\t-8.5501\t-1.3785\tS0\tS1\t0.500\t-1.3735\t-0.0249\t-8.5501\t-0.0199\t-1.3735\t-1.3735\t0.206\t1"
    assert best[2] == "CCC\t3\t3\t3\t2\tSNG-S1\tS1\t-0.0151\tS0\t-9.6990\t-1.3856\tS1\tS0\t0.500\t-1.3856\t-0.0151\t-9.6990\t-1.3856\t-9.6990\t-1.3856\t0.203\t1"


def test_phred_and_store_units(oracle):
    """Rows a2, a1 against the reference's PhredHelper.cpp / sc_drop_seq.cpp compiled alone (ref_units.npz)."""
    z = np.load(GOLDEN / "ref_units.npz")
    mat, err = oracle.phred_tables()
    assert np.array_equal(mat, z["phred_mat"]) and np.array_equal(err, z["phred_err"])
    L = oracle.lib()
    assert np.array_equal(np.array([L.orc_phred_prob(q) for q in range(300)]), z["phred_prob"])
    st = L.orc_store_new()
    rets, ids = [], []
    for c, s, u, a, b in zip(z["ev_cell"], z["ev_snp"], z["ev_umi"], z["ev_allele"], z["ev_bq"]):
        cid = L.orc_store_add_cell(st, str(c).encode())
        ids.append(cid)
        rets.append(L.orc_store_add_read(st, int(s), cid, str(u).encode(), int(a), int(b)))
    assert np.array_equal(np.array(rets), z["ret_new"]) and np.array_equal(np.array(ids), z["ret_cellid"])
    L.orc_store_freeze(st)
    B = L.orc_store_ncells(st)
    import ctypes as C
    def arr(ptr, n, ct): return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()
    P = L.orc_store_npairs(st); R = L.orc_store_nwords(st)
    cell_off = arr(L.orc_store_cell_off(st), B + 1, C.c_int64)
    assert np.array_equal(np.diff(cell_off), z["cell_npairs"])
    assert np.array_equal(arr(L.orc_store_pair_snp(st), P, C.c_int32), z["flat_snp"])
    assert np.array_equal(np.diff(arr(L.orc_store_pair_off(st), P + 1, C.c_int64)), z["flat_nper"])
    assert np.array_equal(arr(L.orc_store_words(st), R, C.c_uint32), z["flat_words"])
    cnt = z["counters"]
    assert np.array_equal(arr(L.orc_store_pass(st), B, C.c_int32), cnt[:, 0])
    assert np.array_equal(arr(L.orc_store_uniq(st), B, C.c_int32), cnt[:, 1])
    L.orc_store_free(st)


def test_oracle_runs_concurrently(oracle):
    """bench.py's all-cores CPU leg calls orc_run from several threads at once: results must equal the serial ones."""
    import threading
    gd = Golden("gt_v64_a2")
    pb = gd.problem(oracle)
    csr = oracle.store_from_events(pb.events)
    want = oracle.run_csr(csr, pb.sample_ids, pb.g, pb.params)
    res = [None] * 6

    def work(i):
        res[i] = oracle.run_csr(csr, pb.sample_ids, pb.g, pb.params)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(res))]
    for t in th: t.start()
    for t in th: t.join()
    for r in res:
        assert np.array_equal(r.llks, want.llks) and np.array_equal(r.llksAB, want.llksAB) and np.array_equal(r.llks00, want.llks00)
