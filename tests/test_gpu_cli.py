"""End to end through the C++ host (ngsdist_amd/bin/ngsDist): same command line
as the reference, byte-identical .dist files.  Expected bytes come from (a) the
reference outputs kept under tests/golden/survey_probe and (b) the CPU oracle
run on the same inputs (oracle.run_reference_flow = the reference's main loop).
"""
import gzip
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ngsdist_amd", "bin", "ngsDist")
SP = os.path.join(ROOT, "tests", "golden", "survey_probe")


def cli(tmp_path, *args, stdin=None, name="out.dist"):
    out = str(tmp_path / name)
    r = subprocess.run([BIN] + [str(a) for a in args] + ["--out", out, "--verbose", "0"], capture_output=True,
                       input=stdin)
    assert r.returncode == 0, r.stderr.decode()
    with open(out) as fh:
        return fh.read()


def golden(name):
    with open(os.path.join(SP, name)) as fh:
        return fh.read()


T_GL = os.path.join(SP, "t_gl.bin")


def test_golden_outputs_of_the_reference(tmp_path):
    assert cli(tmp_path, "--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200, "--indep_geno",
               "--evol_model", 0) == golden("t_gl_I0.dist")
    assert cli(tmp_path, "--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200,
               "--evol_model", 2) == golden("t_gl_EM2.dist")
    assert cli(tmp_path, "--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200, "--call_geno",
               "--evol_model", 0) == golden("t_gl_CG.dist")
    assert cli(tmp_path, "--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200, "--indep_geno", "--n_boot_rep", 2,
               "--boot_block_size", 7, "--seed", 12345) == golden("t_gl_B.dist")
    assert cli(tmp_path, "--geno", os.path.join(SP, "t_geno.gz"), "--n_ind", 6, "--n_sites", 200) == golden("t_T.dist")
    assert cli(tmp_path, "--geno", os.path.join(SP, "id.geno.gz"), "--n_ind", 3, "--n_sites", 4) == golden("id.dist")
    assert cli(tmp_path, "--geno", os.path.join(SP, "id.geno.gz"), "--n_ind", 3, "--n_sites", 4,
               "--pairwise_del") == golden("id2.dist")
    assert cli(tmp_path, "--geno", os.path.join(SP, "far.geno.gz"), "--n_ind", 2, "--n_sites", 2) == golden("far.dist")


def test_every_kernel_prints_the_same_bytes(tmp_path):
    base = ["--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200]
    for k in ("stream", "mfma"):
        assert cli(tmp_path, *base, "--indep_geno", "--evol_model", 0, "--kernel", k) == golden("t_gl_I0.dist")
    for k in ("em_table", "em_fast", "em_faithful"):
        assert cli(tmp_path, *base, "--evol_model", 2, "--kernel", k) == golden("t_gl_EM2.dist")


def test_stdin_binary_and_log_scale(tmp_path):
    raw = np.fromfile(T_GL, dtype=np.float64)
    exp = golden("t_gl_I0.dist")
    assert cli(tmp_path, "--geno", "-", "--probs", "--n_ind", 6, "--n_sites", 200, "--indep_geno", "--evol_model", 0,
               stdin=raw.tobytes()) == exp
    logf = tmp_path / "t_log.bin"
    np.log(raw).tofile(str(logf))
    p = O.prep_binary(np.log(raw), 6, 200, in_logscale=True)
    assert cli(tmp_path, "--geno", logf, "--log_scale", "--n_ind", 6, "--n_sites", 200, "--indep_geno",
               "--evol_model", 0) == O.run_reference_flow(p, evol_model=0)


@pytest.mark.parametrize("flags,kw", [
    (["--call_geno", "--N_thresh", "0.3", "--call_thresh", "0.9"], dict(call=(0.3, 0.9))),
    (["--call_geno", "--pairwise_del", "--N_thresh", "0.5", "--call_thresh", "0.9"], dict(call=(0.5, 0.9), pairwise_del=True)),
    (["--indep_geno", "--avg_nuc_dist", "--evol_model", "2"], dict(avg=True, evol_model=2)),
    (["--indep_geno", "--tot_sites", "1000", "--evol_model", "0"], dict(tot_sites=1000, evol_model=0)),
    (["--pairwise_del", "--evol_model", "1"], dict(pairwise_del=True, indep_geno=False)),
    (["--indep_geno", "--n_boot_rep", "3", "--boot_block_size", "1", "--seed", "99"], dict(n_boot_rep=3, boot_block_size=1, seed=99)),
    (["--n_boot_rep", "2", "--boot_block_size", "16", "--seed", "5", "--evol_model", "2"],
     dict(n_boot_rep=2, boot_block_size=16, seed=5, indep_geno=False, evol_model=2)),
    # more replicates than one engine batch (32), blocks that leave a tail of sites, per-block partial (sum, cnt)
    (["--indep_geno", "--pairwise_del", "--n_boot_rep", "40", "--boot_block_size", "12", "--seed", "7", "--n_threads", "3"],
     dict(n_boot_rep=40, boot_block_size=12, seed=7, pairwise_del=True)),
    # EM path: the whole job (full data + 40 replicates) goes to the engine in ONE call -- the per-site EM serves them all
    (["--n_boot_rep", "40", "--boot_block_size", "3", "--seed", "11", "--evol_model", "2"],
     dict(n_boot_rep=40, boot_block_size=3, seed=11, indep_geno=False, evol_model=2)),
])
def test_flag_combinations_against_oracle_flow(tmp_path, flags, kw):
    raw = np.fromfile(T_GL, dtype=np.float64)
    call = kw.pop("call", None)
    avg = kw.pop("avg", False)
    p = O.prep_binary(raw, 6, 200, call_geno=call is not None, N_thresh=call[0] if call else 0,
                      call_thresh=call[1] if call else 0)
    kw.setdefault("indep_geno", True)
    exp = O.run_reference_flow(p, score=O.score_matrix(avg), **kw)
    got = cli(tmp_path, "--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200, *flags)
    assert got == exp


def _testA_like(tmp_path, n_ind=24, n_sites=10000, seed=4):
    """examples/test.sh testA shape: called genotypes, gz TSV with two prefix columns,
    labels with awkward characters (examples/testA.labels style)."""
    rng = np.random.default_rng(seed)
    freq = rng.uniform(0.05, 0.5, size=n_sites)
    g = (rng.random((n_sites, n_ind, 2)) < freq[:, None, None]).sum(axis=2)
    g[rng.random((n_sites, n_ind)) < 0.02] = -1
    path = tmp_path / "testA_T.geno.gz"
    with gzip.open(str(path), "wt") as fh:
        for s in range(n_sites):
            fh.write("chrSIM\t%d\t" % (s + 1) + "\t".join(str(int(x)) for x in g[s]) + "\n")
    labels = ["pop%d_ind%d%s" % (i // 8, i, "*+#"[i % 3]) + ("\textra" if i % 5 == 0 else "") for i in range(n_ind)]
    lpath = tmp_path / "testA.labels"
    lpath.write_text("\n".join(labels) + "\n")
    return str(path), str(lpath), [l.split("\t")[0] for l in labels]


def test_testA_shape_called_genotypes(tmp_path):
    path, lpath, labels = _testA_like(tmp_path)
    p = O.load_text(path, 24, 10000, in_probs=False)
    for extra, kw in ((["--n_threads", "10", "--seed", "12345"], {}),
                      (["--seed", "12345", "--n_boot_rep", "5"], dict(n_boot_rep=5)),
                      (["--seed", "12345", "--n_boot_rep", "5", "--boot_block_size", "10"],
                       dict(n_boot_rep=5, boot_block_size=10))):
        exp = O.run_reference_flow(p, labels=labels, seed=12345, n_threads=8, **kw)
        got = cli(tmp_path, "--geno", path, "--n_ind", 24, "--n_sites", 10000, "--labels", lpath, *extra)
        assert hashlib.md5(got.encode()).hexdigest() == hashlib.md5(exp.encode()).hexdigest()
        assert got == exp


def test_the_references_test_script_matrix_on_likelihood_inputs(tmp_path):
    """examples/test.sh beyond its three called-genotype runs: the same 24 x 10 000 shape as genotype LIKELIHOODS in the
    script's three encodings -- BEAGLE text (header line, marker + two allele columns, --pos file), binary doubles,
    posterior-probability text (two prefix columns) -- each through its five command lines: plain, --n_boot_rep 5 at the
    default block size 1 and at --boot_block_size 10 (the EM path, the reference's default), the latter with
    --call_geno and with --call_geno --N_thresh 0.3 --call_thresh 0.9; always --n_threads 10 --seed 12345 --labels.  The
    script's own inputs come from ngsSim + ANGSD (absent), so the data are synthetic and the expected text is the
    oracle's restatement of the reference's main loop: same bytes, or (likelihood sums differ in their last bits)
    every cell within 2e-10."""
    n_ind, n_sites = 24, 10000
    _, lpath, labels = _testA_like(tmp_path, n_sites=10)  # the label file (and a small genotype file not used here)
    p = O.synth_indmajor(17, n_ind, n_sites, miss_frac=0.03)
    sm = np.ascontiguousarray(p.transpose(1, 0, 2))  # [site][ind][3], the file order
    beagle, binf, post, pos = (str(tmp_path / n) for n in ("testA_2.beagle.gz", "testA_32.geno", "testA_8.geno.gz", "testA.pos"))
    fmt = lambda row: "\t".join("%.17g" % x for x in row.reshape(-1))
    with gzip.open(beagle, "wt") as fh:
        fh.write("marker\tallele1\tallele2\t" + "\t".join("Ind%d\tInd%d\tInd%d" % (i, i, i) for i in range(n_ind)) + "\n")
        for k in range(n_sites):
            fh.write("chrSIM_%d\t%d\t%d\t" % (k + 1, k % 4, (k + 1) % 4) + fmt(sm[k]) + "\n")
    with open(pos, "w") as fh:
        fh.write("".join("chrSIM\t%d\n" % (k + 1) for k in range(n_sites)))
    sm.tofile(binf)
    with gzip.open(post, "wt") as fh:
        for k in range(n_sites):
            fh.write("chrSIM\t%d\t" % (k + 1) + fmt(sm[k]) + "\n")
    runs = [([], {}), (["--n_boot_rep", 5], dict(n_boot_rep=5)),
            (["--n_boot_rep", 5, "--boot_block_size", 10], dict(n_boot_rep=5, boot_block_size=10)),
            (["--n_boot_rep", 5, "--boot_block_size", 10, "--call_geno"], dict(n_boot_rep=5, boot_block_size=10, call=(0.0, 0.0))),
            (["--n_boot_rep", 5, "--boot_block_size", 10, "--call_geno", "--N_thresh", 0.3, "--call_thresh", 0.9],
             dict(n_boot_rep=5, boot_block_size=10, call=(0.3, 0.9)))]
    same = 0
    for path, extra_in, text in ((beagle, ["--pos", pos], True), (binf, [], False), (post, [], True)):
        prepared = {}
        for flags, kw in runs:
            kw = dict(kw)
            call = kw.pop("call", None)
            if call not in prepared:  # the reference prepares (and calls) at load time: once per calling rule
                ck = dict(call_geno=call is not None, N_thresh=call[0] if call else 0.0, call_thresh=call[1] if call else 0.0)
                prepared[call] = O.load_text(path, n_ind, n_sites, in_probs=True, **ck) if text else \
                    O.prep_binary(sm, n_ind, n_sites, **ck)
            exp = O.run_reference_flow(prepared[call], labels=labels, seed=12345, n_threads=16,
                                       indep_geno=call is not None, **kw)  # --call_geno forces --indep_geno (ngsDist.cpp:55-65)
            got = cli(tmp_path, "--geno", path, "--probs", "--n_ind", n_ind, "--n_sites", n_sites, "--labels", lpath,
                      "--n_threads", 10, "--seed", 12345, *extra_in, *flags)
            if got == exp:
                same += 1
            else:
                a, b = cells(got), cells(exp)
                assert a.shape == b.shape and [ln.split("\t")[0] for ln in got.splitlines()] == [ln.split("\t")[0] for ln in exp.splitlines()]
                assert np.all((np.abs(a - b) <= 2e-10 * np.maximum(1.0, np.abs(b))) | (a == b) | (np.isnan(a) & np.isnan(b))), (path, flags)
    assert same >= 6  # the called-genotype runs are dyadic sums: identical bytes whatever the order of additions


def test_text_gl_with_header_and_two_gpu_shards_on_one_device(tmp_path):
    n_ind, n_sites = 150, 300  # two tile rows -> 3 pair tiles
    p_raw = O.synth_indmajor(31, n_ind, n_sites)  # [i][s][3]
    path = tmp_path / "gl.beagle.gz"
    with gzip.open(str(path), "wt") as fh:
        fh.write("marker\tallele1\tallele2\t" + "\t".join("Ind%d" % (i // 3) for i in range(3 * n_ind)) + "\n")
        for s in range(n_sites):
            fh.write("chr_%d\t0\t1\t" % s + "\t".join("%.17g" % v for v in p_raw[:, s, :].reshape(-1)) + "\n")
    p = O.load_text(str(path), n_ind, n_sites, in_probs=True)
    exp = O.run_reference_flow(p, evol_model=1, indep_geno=True, n_threads=8)
    got = cli(tmp_path, "--geno", path, "--probs", "--n_ind", n_ind, "--n_sites", n_sites, "--indep_geno")
    assert got == exp


def test_prep_on_device_prints_the_same_bytes(tmp_path):
    base = ["--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200]
    for prep in ("host", "device"):
        assert cli(tmp_path, *base, "--indep_geno", "--evol_model", 0, "--prep", prep) == golden("t_gl_I0.dist")
        assert cli(tmp_path, *base, "--evol_model", 2, "--prep", prep) == golden("t_gl_EM2.dist")
        assert cli(tmp_path, *base, "--call_geno", "--evol_model", 0, "--prep", prep) == golden("t_gl_CG.dist")


def test_nan_in_binary_input_is_the_reference_error(tmp_path):
    raw = np.fromfile(T_GL, dtype=np.float64).copy()
    raw[100] = -1.0
    bad = tmp_path / "bad.bin"
    raw.tofile(str(bad))
    for prep in ("host", "device"):
        r = subprocess.run([BIN, "--geno", str(bad), "--probs", "--n_ind", "6", "--n_sites", "200", "--out",
                            str(tmp_path / "o"), "--verbose", "0", "--prep", prep], capture_output=True)
        assert r.returncode == 255 and b"NaN found! Is the file format correct?" in r.stderr


def test_text_input_edges_empty_line_header_extra_columns(tmp_path):
    """read_data.cpp:48-103: an empty line keeps the site at its fill, a short first line is a header,
    non-numeric fields are dropped, only the LAST n_ind*n_geno columns are used."""
    n_ind, n_sites = 5, 12
    rng = np.random.default_rng(2)
    g = rng.integers(-1, 3, size=(n_sites, n_ind))
    path = tmp_path / "edges.geno.gz"
    with gzip.open(str(path), "wt") as fh:
        fh.write("chr\tpos\n")  # header: fewer numeric fields than n_ind
        for s in range(n_sites):
            if s == 4:
                fh.write("\n")  # empty line
            else:
                fh.write("chrX\t%d\tNA\t7\t" % s + " ".join(str(int(x)) for x in g[s]) + "\n")
    p = O.load_text(str(path), n_ind, n_sites, in_probs=False)
    assert np.all(p[:, 4, :] == 0.0)
    for extra, kw in (([], {}), (["--pairwise_del"], dict(pairwise_del=True)), (["--evol_model", "0"], dict(evol_model=0))):
        exp = O.run_reference_flow(p, **kw)
        assert cli(tmp_path, "--geno", path, "--n_ind", n_ind, "--n_sites", n_sites, *extra) == exp


def test_text_input_errors(tmp_path):
    path = tmp_path / "short.geno.gz"
    with gzip.open(str(path), "wt") as fh:
        fh.write("0\t1\t2\n0\t1\n")  # second line: less fields than expected
    r = subprocess.run([BIN, "--geno", str(path), "--n_ind", "3", "--n_sites", "2", "--out", str(tmp_path / "o"),
                        "--verbose", "0"], capture_output=True)
    assert r.returncode == 255 and b"Less fields than expected!" in r.stderr
    with gzip.open(str(path), "wt") as fh:
        fh.write("0\t1\t2\n")
    r = subprocess.run([BIN, "--geno", str(path), "--n_ind", "3", "--n_sites", "2", "--out", str(tmp_path / "o"),
                        "--verbose", "0"], capture_output=True)
    assert r.returncode == 255 and b"premature EOF" in r.stderr
    with gzip.open(str(path), "wt") as fh:
        fh.write("0\t1\t2\n0\t1\t2\n0\t1\t2\n")
    r = subprocess.run([BIN, "--geno", str(path), "--n_ind", "3", "--n_sites", "2", "--out", str(tmp_path / "o"),
                        "--verbose", "0"], capture_output=True)
    assert r.returncode == 255 and b"not at EOF" in r.stderr
    with gzip.open(str(path), "wt") as fh:
        fh.write("0\t1\t3\n0\t1\t2\n")
    r = subprocess.run([BIN, "--geno", str(path), "--n_ind", "3", "--n_sites", "2", "--out", str(tmp_path / "o"),
                        "--verbose", "0"], capture_output=True)
    assert r.returncode == 255 and b"Genotypes must be coded as {-1,0,1,2}" in r.stderr


def cells(text):
    return np.array([float(x) for ln in text.splitlines() if "\t" in ln for x in ln.split("\t")[1:]])


def test_multi_gpu_site_ranges_on_one_device(tmp_path):
    """--n_gpus 3 --same_device: the site axis in three ranges, each read, held and computed by its own engine and
    host thread, sums added.  Called genotypes (every term dyadic): the same bytes as one engine -- through the
    stream reader (gz text: ranges are read front to back and computed while the next is read) and through the
    binary file (every device's thread reads its own range), bootstrap and --pairwise_del included.  GL data
    (indep and EM paths): equal to 1e-9."""
    path, lpath, labels = _testA_like(tmp_path)
    base = ["--geno", path, "--n_ind", 24, "--n_sites", 10000, "--labels", lpath, "--seed", 12345]
    for extra in ([], ["--n_boot_rep", 3, "--boot_block_size", 10, "--pairwise_del"], ["--n_boot_rep", 2, "--n_threads", 3]):
        one = cli(tmp_path, *base, *extra, name="one.dist")
        r = subprocess.run([BIN] + [str(a) for a in base + extra] + ["--n_gpus", "3", "--same_device", "--out",
                                                                      str(tmp_path / "three.dist"), "--verbose", "1"],
                           capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        assert b"Site axis split over 3 devices: 3 ranges" in r.stderr
        assert open(str(tmp_path / "three.dist")).read() == one
    n_ind, n_sites = 300, 2000
    raw = O.synth_indmajor(9, n_ind, n_sites, miss_frac=0.1).transpose(1, 0, 2).copy()
    gl = tmp_path / "g.bin"
    raw.tofile(str(gl))
    base = ["--geno", gl, "--probs", "--n_ind", n_ind, "--n_sites", n_sites]
    # called on the host from the binary GLs: dyadic terms again -> identical bytes from parallel range readers
    called = ["--call_geno", "--n_boot_rep", "2", "--boot_block_size", "20", "--seed", "3"]
    assert cli(tmp_path, *base, *called, "--n_gpus", 4, "--same_device", name="four.dist") == cli(tmp_path, *base, *called)
    for extra in (["--indep_geno", "--pairwise_del"], ["--evol_model", "2"],
                  ["--indep_geno", "--n_boot_rep", "2", "--boot_block_size", "20", "--seed", "3"]):
        a = cells(cli(tmp_path, *base, *extra, name="one.dist"))
        b = cells(cli(tmp_path, *base, *extra, "--n_gpus", 3, "--same_device", name="three.dist"))
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-9, atol=1e-12)


def test_single_image_flag_prints_the_same_bytes(tmp_path):
    """--single_image (ngd_config.single_image = 2: one operand image in coordinates in which the score matrix is
    diagonal): called genotypes print the same bytes (bootstrap, --pairwise_del, --avg_nuc_dist, site ranges on several
    engines included); GL data equal to 1e-9; on the EM path the flag changes nothing"""
    path, lpath, labels = _testA_like(tmp_path)
    base = ["--geno", path, "--n_ind", 24, "--n_sites", 10000, "--labels", lpath, "--seed", 12345, "--indep_geno"]
    for extra in ([], ["--n_boot_rep", 3, "--boot_block_size", 10, "--pairwise_del"], ["--avg_nuc_dist", "--evol_model", 0],
                  ["--n_boot_rep", 2, "--n_gpus", 2, "--same_device"]):
        assert cli(tmp_path, *base, *extra, "--single_image", name="single.dist") == cli(tmp_path, *base, *extra, name="two.dist")
    n_ind, n_sites = 300, 2000
    raw = O.synth_indmajor(9, n_ind, n_sites, miss_frac=0.1).transpose(1, 0, 2).copy()
    gl = tmp_path / "g.bin"
    raw.tofile(str(gl))
    base = ["--geno", gl, "--probs", "--n_ind", n_ind, "--n_sites", n_sites]
    for extra in (["--indep_geno", "--pairwise_del"], ["--indep_geno", "--n_boot_rep", "2", "--boot_block_size", "20", "--seed", "3"]):
        a = cells(cli(tmp_path, *base, *extra, name="two.dist"))
        b = cells(cli(tmp_path, *base, *extra, "--single_image", name="single.dist"))
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-9, atol=1e-12)
    assert cli(tmp_path, *base, "--single_image", name="single.dist") == cli(tmp_path, *base, name="two.dist")  # EM path


def test_fewer_sites_than_one_bootstrap_block(tmp_path):
    """--n_boot_rep with n_sites < --boot_block_size: the reference truncates the replicates to 0 sites
    (ngsDist.cpp:236), visits none and prints 0/0 cells after the full-data matrix"""
    raw = np.fromfile(T_GL, dtype=np.float64)
    p = O.prep_binary(raw, 6, 200)
    exp = O.run_reference_flow(p, evol_model=1, indep_geno=True, n_boot_rep=2, boot_block_size=500, seed=1)
    assert "nan" in exp
    for extra in ([], ["--n_gpus", 2, "--same_device"]):
        assert cli(tmp_path, "--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200, "--indep_geno", "--n_boot_rep", 2,
                   "--boot_block_size", 500, "--seed", 1, *extra) == exp


def test_block_size_whose_lcm_with_16_exceeds_the_data_set(tmp_path):
    """--n_gpus ranges are whole multiples of lcm(16, --boot_block_size); a large odd block size makes that exceed
    n_sites (lcm(16, 67) = 1072 > 200): the job then runs as ONE range on one device instead of dying for want of a
    device that holds 1072 sites -- same bytes as the one-engine run"""
    base = ["--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200, "--indep_geno", "--n_boot_rep", 3,
            "--boot_block_size", 67, "--seed", 5]
    exp = cli(tmp_path, *base)
    assert cli(tmp_path, *base, "--n_gpus", 2, "--same_device", name="two.dist") == exp
    assert cli(tmp_path, *base, "--n_gpus", 2, "--same_device", "--max_device_bytes", 700 << 20, name="cap.dist") == exp


def test_bench_line_contract(tmp_path):
    """bench.py prints ONE JSON line with the fields the driver reads."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--n_sites", "4000",
                        "--steps", "2", "--warmup", "1", "--cpu_sites", "400"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port"
    assert d["spot_check"]["max_rel_err_vs_oracle"] < 1e-9


def test_bench_pipelined_tail_and_em_roofline_fields(tmp_path):
    """N = 1: the headline region runs one job at a time (ms_per_step = one job's latency); a second region runs the same
    jobs two deep -- a job's copy-out and host tail on the worker thread beside the next job's kernels (forced here on
    small jobs: --pipelined_tail) -- and is reported apart as `pipelined`.  With --vary_jobs odd and even steps compute
    different jobs and every tail records a checksum of its whole result: the pipelined region must reproduce the serial
    region's checksums step by step, so a mix-up of the two buffer sets cannot pass.  The EM line carries the
    active-lane and issue-slot figures (or says why not) and the reference-em2 CPU baseline."""
    import json
    import sys

    def line(*args):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1"] + list(args),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])

    for extra in (["--workload", "cfg2", "--n_sites", "20000"], ["--workload", "cfg5", "--n_sites", "20000"]):
        a = line(*extra, "--no_cpu", "--pipelined_tail", "--vary_jobs")
        b = line(*extra, "--no_cpu", "--serial_tail")
        assert a["valid"] and b["valid"] and a["config"]["host_tail"].startswith("serial")
        assert a["pipelined"] is not None and a["pipelined"]["ms_per_step"] > 0 and b["pipelined"] is None
        assert a["pipeline_check"] == {"steps_compared": 5, "jobs_differ": True, "ok": True}
        assert a["spot_check"]["max_rel_err_vs_oracle"] == b["spot_check"]["max_rel_err_vs_oracle"]
        assert a["roofline"]["launches_timed"] >= 5 and a["roofline"]["ms_per_launch_min"] <= a["roofline"]["ms_per_launch_median"]
    d = line("--workload", "cfg4", "--n_sites", "3000", "--cpu_sites", "60")
    assert d["valid"] and d["roofline"]["kernel"] == "k_accum_em_table" and "frac_kind" in d["roofline"]
    assert d["roofline"]["bound"] == "valu" and d["roofline"]["unit"] == "lane-instructions/s"
    if d["roofline"]["frac"] is not None:
        assert d["roofline"]["frac"] == d["roofline"]["active_lane_frac"] <= d["roofline"]["issue_slot_frac"]
    ref = d["cpu_baseline"]["reference_em2"]
    assert ref.get("kind") == "reference-em2" and ref["bit_identical_to_port"] is True, ref


def test_binary_input_size_errors_and_gz_binary(tmp_path):
    """plain binary files are read with pread on several threads, gz-compressed binary ones (and stdin) through
    gzread: same bytes out; a file with trailing bytes is the reference's "not at EOF", a wrong size its
    "invalid/corrupt" error."""
    raw = np.fromfile(T_GL, dtype=np.float64)
    base = ["--probs", "--n_ind", 6, "--n_sites", 200, "--indep_geno"]
    exp = cli(tmp_path, "--geno", T_GL, *base)
    for prep in ("host", "device"):
        assert cli(tmp_path, "--geno", T_GL, *base, "--prep", prep, "--n_threads", 5) == exp
    # trailing bytes (less than one site): passes the size check like the reference's integer division, then fails
    tail = tmp_path / "tail.bin"
    with open(str(tail), "wb") as fh:
        fh.write(raw.tobytes() + b"\0" * 8)
    for prep in ("host", "device"):
        r = subprocess.run([BIN, "--geno", str(tail), "--probs", "--n_ind", "6", "--n_sites", "200", "--out",
                            str(tmp_path / "o"), "--verbose", "0", "--prep", prep], capture_output=True)
        assert r.returncode == 255 and b"not at EOF" in r.stderr
    short = tmp_path / "short.bin"
    raw[:-18].tofile(str(short))
    r = subprocess.run([BIN, "--geno", str(short), "--probs", "--n_ind", "6", "--n_sites", "200", "--out",
                        str(tmp_path / "o"), "--verbose", "0"], capture_output=True)
    assert r.returncode == 255 and b"invalid/corrupt genotype input file" in r.stderr
    # binary through stdin (gzread path)
    with open(T_GL, "rb") as fh:
        got = cli(tmp_path, "--geno", "-", *base, stdin=fh.read())
    assert got == exp


def test_data_set_larger_than_the_device_goes_through_in_ranges(tmp_path):
    """--max_device_bytes below the data set's footprint: the host sends ranges of sites through the device one
    after the other and adds the per-range (sum, cnt).  Called genotypes (every term dyadic): byte-identical to the
    one-engine run, bootstrap replicates, --pairwise_del and text input included; GL data on the EM path: equal to 1e-9."""
    path, lpath, labels = _testA_like(tmp_path)
    base = ["--geno", path, "--n_ind", 24, "--n_sites", 10000, "--labels", lpath, "--seed", 12345]
    # footprint model of the host: 512 MiB + slabs + 64 B per pair fixed, ~6.2 KB per site at 24 individuals
    small = str((512 << 20) + 256 * 128 * 128 * 8 + 276 * 64 + 6200 * 2600)
    for extra in ([], ["--n_boot_rep", 3, "--boot_block_size", 10, "--pairwise_del"], ["--n_boot_rep", 2, "--n_threads", 3]):
        whole = cli(tmp_path, *base, *extra)
        r = subprocess.run([BIN] + [str(a) for a in base + extra] + ["--max_device_bytes", small, "--out",
                                                                      str(tmp_path / "parts.dist"), "--verbose", "1"],
                           capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        assert b"larger than the device budget" in r.stderr and b" 4 ranges" in r.stderr
        assert open(str(tmp_path / "parts.dist")).read() == whole
    # binary GL input, EM path, bootstrap with 16-site blocks: numerically equal; also two devices that are too small
    gl = ["--geno", T_GL, "--probs", "--n_ind", 6, "--n_sites", 200, "--n_boot_rep", 2, "--boot_block_size", 16, "--seed", 5]
    whole = cli(tmp_path, *gl)
    tiny = str((512 << 20) + 256 * 128 * 128 * 8 + 15 * 64 + 3200 * 70)
    for extra in ([], ["--n_gpus", 2, "--same_device"]):
        parts = cli(tmp_path, *gl, "--max_device_bytes", tiny, *extra, name="p2.dist")
        a, b = cells(whole), cells(parts)
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-9, atol=1e-10)


def _write_bgzf(path, data, block=0xff00):
    """blocked gzip as bgzip / htslib / ANGSD write it"""
    import struct
    import zlib
    with open(path, "wb") as fh:
        for k in list(range(0, len(data), block)) + [None]:
            chunk = b"" if k is None else data[k:k + block]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = c.compress(chunk) + c.flush()
            fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                     + struct.pack("<II", zlib.crc32(chunk), len(chunk)))


def test_bgzf_text_is_inflated_on_several_threads_and_prints_the_same_bytes(tmp_path):
    """a .gz that is BGZF (independent members of <= 64 KB) goes through the block-parallel reader when --n_threads > 1;
    everything downstream is the same code: same bytes as the plain .gz and as the oracle's flow, lines that straddle
    blocks, the reference's EOF errors"""
    path, lpath, labels = _testA_like(tmp_path)
    data = gzip.open(path, "rb").read()
    p = O.load_text(path, 24, 10000, in_probs=False)
    exp = O.run_reference_flow(p, labels=labels, seed=12345, n_threads=8, n_boot_rep=2, boot_block_size=10)
    base = ["--n_ind", 24, "--n_sites", 10000, "--labels", lpath, "--seed", 12345, "--n_boot_rep", 2, "--boot_block_size", 10]
    for block in (0xff00, 777):
        bz = str(tmp_path / ("b%d.geno.gz" % block))
        _write_bgzf(bz, data, block)
        for thr in (1, 5, 16):
            r = subprocess.run([BIN, "--geno", bz, "--n_threads", str(thr), "--out", str(tmp_path / "b.dist"), "--verbose", "2"]
                               + [str(a) for a in base], capture_output=True)
            assert r.returncode == 0, r.stderr.decode()
            assert (b"BGZF input" in r.stderr) == (thr > 1)
            assert open(str(tmp_path / "b.dist")).read() == exp
    r = subprocess.run([BIN, "--geno", bz, "--n_threads", "4", "--n_ind", "24", "--n_sites", "10001", "--out", str(tmp_path / "x")],
                       capture_output=True)
    assert r.returncode != 0 and b"premature EOF" in r.stderr
    r = subprocess.run([BIN, "--geno", bz, "--n_threads", "4", "--n_ind", "24", "--n_sites", "9999", "--out", str(tmp_path / "x")],
                       capture_output=True)
    assert r.returncode != 0 and b"not at EOF" in r.stderr


def test_end_to_end_bench_tool_checks_every_printed_cell(tmp_path):
    """tools/bench_e2e.py on cfg 2 (200 x 1e5, a 0.48 GB file): the host runs from a generated file through the mapped-file
    loader (ring of pinned buffers, device memory in pieces), reports its phases, and every printed cell agrees with an
    engine filled on the device from the same seed."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_e2e.py"), "--workloads", "cfg2", "--runs", "2", "--gap", "0",
                        "--no_roof", "--dir", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["valid"] is True and line["check"]["cells"] == 19900
    ph = line["phases_s"]
    assert ph["load"] > 0 and ph["total_since_main"] <= line["wall_s"] and "of_load_read" in ph
