"""The C++ host keeps the reference's argument checks (parse_args.cpp:203-220)
and its error convention (gen_func.cpp:12-18: message on stderr, exit(-1)).
These paths end before any GPU work, so they run on CPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ngsdist_amd", "bin", "ngsDist")
SP = os.path.join(ROOT, "tests", "golden", "survey_probe")

pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="host binary not built")


def run(*args):
    return subprocess.run([BIN] + list(args), capture_output=True, text=True)


@pytest.mark.parametrize("args,msg", [
    ([], "genotype input file (--geno) missing!"),
    (["--geno", "x"], "number of individuals (--n_ind) missing!"),
    (["--geno", "x", "--n_ind", "3"], "number of sites (--n_sites) missing!"),
    (["--geno", "x", "--n_ind", "3", "--n_sites", "4", "--tot_sites", "9", "--pairwise_del", "--out", "o"],
     "cannot specify total number of sites (--tot_sites) with pairwise deletion (--pairwise_del)!"),
    (["--geno", "x", "--n_ind", "3", "--n_sites", "4", "--call_geno", "--out", "o"],
     "can only call genotypes from likelihoods/probabilities!"),
    (["--geno", "x", "--n_ind", "3", "--n_sites", "4", "--evol_model", "9", "--out", "o"],
     "invalid correction method specified!"),
    (["--geno", "x", "--n_ind", "3", "--n_sites", "4", "--evol_model", "3", "--out", "o"],
     "use of more complex evolutionary models requires position information!"),
    (["--geno", "x", "--n_ind", "3", "--n_sites", "4"], "output prefix (--out) missing!"),
    (["--geno", "x", "--n_ind", "3", "--n_sites", "4", "--out", "o", "--n_threads", "0"],
     "number of threads cannot be less than 1!"),
])
def test_argument_checks(args, msg):
    r = run(*args, "--verbose", "0")
    assert r.returncode == 255  # exit(-1)
    assert "ERROR: [parse_cmd_args] " + msg in r.stderr


def test_unknown_option_exits_minus_one():
    r = run("--no_such_flag")
    assert r.returncode == 255


def test_missing_input_file(tmp_path):
    r = run("--geno", str(tmp_path / "nope.gz"), "--n_ind", "3", "--n_sites", "4", "--out", str(tmp_path / "o"),
            "--verbose", "0")
    assert r.returncode == 255 and "cannot check GENO file size!" in r.stderr


def test_binary_size_check(tmp_path):
    r = run("--geno", os.path.join(SP, "t_gl.bin"), "--n_ind", "6", "--n_sites", "199", "--out",
            str(tmp_path / "o"), "--verbose", "0")
    assert r.returncode == 255 and "invalid/corrupt genotype input file!" in r.stderr


def test_single_dash_long_options_are_accepted(tmp_path):
    # getopt_long_only: "-n_ind 3" works like "--n_ind 3" (parse_args.cpp:83)
    r = run("-geno", "x", "-n_ind", "3", "-verbose", "0")
    assert "number of sites (--n_sites) missing!" in r.stderr


def test_gpu_options_are_checked_before_any_gpu_work(tmp_path):
    r = run("--geno", "x", "--n_ind", "3", "--n_sites", "4", "--out", "o", "--n_gpus", "0", "--verbose", "0")
    assert r.returncode == 255 and "number of GPUs cannot be less than 1!" in r.stderr
    r = run("--geno", "x", "--n_ind", "3", "--n_sites", "4", "--out", "o", "--kernel", "no_such_kernel", "--verbose", "0")
    assert r.returncode == 255


def test_without_a_device_the_host_says_so_instead_of_falling_back(tmp_path):
    import ngsdist_amd as N
    if N.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = run("--geno", os.path.join(SP, "t_gl.bin"), "--probs", "--n_ind", "6", "--n_sites", "200", "--out",
            str(tmp_path / "o"), "--verbose", "0", "--n_gpus", "2", "--same_device")
    assert r.returncode == 255 and "no HIP device found (this program has no CPU path)" in r.stderr
    assert not os.path.exists(str(tmp_path / "o")) or os.path.getsize(str(tmp_path / "o")) == 0
