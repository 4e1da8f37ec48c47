"""The C++ host under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on the pool): it is compiled
against tests/host_sanitize/stub_engine.cpp -- the device entry points with no compute behind them -- and driven through
its readers, the pinned-buffer pipeline, site ranges, bootstrap bookkeeping and printing.  Checked: no sanitizer report,
the exit codes, the shape of what is printed.  (What the numbers are is the GPU suite's business.)"""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ngsdist_amd", "csrc", "host", "ngsdist_host.cpp")


@pytest.fixture(scope="module")
def san_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("san") / "ngsDist_san")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-pthread", "-o", out, HOST,
           os.path.join(ROOT, "tests", "host_sanitize", "stub_engine.cpp"),
           os.path.join(ROOT, "ngsdist_amd", "csrc", "host_util.cpp"), "-I" + os.path.join(ROOT, "ngsdist_amd", "csrc"), "-lz"]
    r = subprocess.run(cmd, capture_output=True)
    if r.returncode != 0:
        err = r.stderr.decode()
        # only a missing sanitizer runtime is a reason to skip; anything else (the stub lagging behind the header, say) fails
        if "libasan" in err or "libubsan" in err or "unrecognized" in err and "fsanitize" in err:
            pytest.skip("no sanitizer runtime here: " + err[-300:])
        pytest.fail("the host does not build against the stub engine:\n" + err[-2000:])
    return out


def run(san_bin, tmp_path, args, stdin=None, ok=True):
    out = str(tmp_path / "o.dist")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([san_bin] + [str(a) for a in args] + ["--out", out, "--verbose", "1"], capture_output=True,
                       stdin=stdin, env=env, timeout=300)
    err = r.stderr.decode(errors="replace")
    assert "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    assert (r.returncode == 0) == ok, err[-1500:]
    return open(out).read() if ok else err


def write_bgzf(path, data, block=0xff00, level=6):
    """blocked gzip as bgzip / htslib write it: members of at most 64 KB with their size in a 'BC' extra field"""
    import struct
    import zlib
    with open(path, "wb") as fh:
        for k in list(range(0, len(data), block)) + [None]:
            chunk = b"" if k is None else data[k:k + block]  # (None: the empty end-of-file block)
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = c.compress(chunk) + c.flush()
            fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25)
                     + comp + struct.pack("<II", zlib.crc32(chunk), len(chunk)))


def n_rows(text):
    return sum(1 for l in text.split("\n") if l.count("\t") > 1)


def test_readers_ranges_bootstrap_and_printing_under_sanitizers(san_bin, tmp_path):
    rng = np.random.default_rng(1)
    case = 0
    for n_ind, n_sites in ((2, 1), (6, 200), (24, 1000), (65, 777)):
        raw = rng.dirichlet([0.5, 0.5, 0.5], size=(n_sites, n_ind))
        gl = str(tmp_path / ("g%d.bin" % case))
        raw.tofile(gl)
        txt = str(tmp_path / ("t%d.gz" % case))
        with gzip.open(txt, "wt") as fh:
            fh.write("chr\tpos\t" + "\t".join("i%d" % i for i in range(n_ind)) + "\n")
            for s in range(n_sites):
                if s == 3 and n_sites > 10:
                    fh.write("\n")  # an empty line is a site of missing data
                    continue
                fh.write("c\t%d\t" % s + "\t".join(str(int(x)) for x in rng.integers(-1, 3, size=n_ind)) + "\n")
        gtxt = str(tmp_path / ("p%d.gz" % case))
        with gzip.open(gtxt, "wt") as fh:
            for s in range(n_sites):
                fh.write("\t".join("%.6g" % x for x in raw[s].reshape(-1)) + "  \textra\n")
        labels = str(tmp_path / ("l%d.txt" % case))
        open(labels, "w").write("".join("lab_%d\tx\n" % i for i in range(n_ind)))
        base_b = ["--geno", gl, "--probs", "--n_ind", n_ind, "--n_sites", n_sites]
        B = max(1, min(10, n_sites))
        for extra in ([], ["--indep_geno", "--evol_model", 0], ["--call_geno", "--N_thresh", 0.3, "--call_thresh", 0.9],
                      ["--log_scale", "--pairwise_del"], ["--n_boot_rep", 3, "--boot_block_size", B, "--seed", 5],
                      ["--n_boot_rep", 40, "--boot_block_size", 1, "--n_threads", 7, "--labels", labels],
                      ["--prep", "host"], ["--prep", "device", "--avg_nuc_dist"],
                      ["--n_gpus", 3, "--n_boot_rep", 2, "--boot_block_size", B],
                      ["--n_gpus", 2, "--same_device", "--pairwise_del", "--n_boot_rep", 2, "--boot_block_size", B],
                      ["--max_device_bytes", (512 << 20) + (1 << 26) + 300 * n_ind * max(64, n_sites // 3), "--n_boot_rep", 2,
                       "--boot_block_size", B]):
            t = run(san_bin, tmp_path, base_b + extra)
            n_mat = 1 + (int(extra[extra.index("--n_boot_rep") + 1]) if "--n_boot_rep" in extra else 0)
            assert n_rows(t) == n_mat * n_ind or n_ind == 2, (extra, n_rows(t))
        run(san_bin, tmp_path, ["--geno", "-", "--probs", "--n_ind", n_ind, "--n_sites", n_sites], stdin=open(gl, "rb"))
        for extra in ([], ["--n_boot_rep", 2, "--boot_block_size", B, "--n_threads", 4], ["--n_gpus", 2, "--pairwise_del"]):
            run(san_bin, tmp_path, ["--geno", txt, "--n_ind", n_ind, "--n_sites", n_sites] + extra)
            run(san_bin, tmp_path, ["--geno", gtxt, "--probs", "--n_ind", n_ind, "--n_sites", n_sites] + extra)
        # error exits: too few sites in the file, too many, a corrupt size, NaN in the data, a short line
        run(san_bin, tmp_path, ["--geno", gl, "--probs", "--n_ind", n_ind, "--n_sites", n_sites + 1], ok=False)
        run(san_bin, tmp_path, ["--geno", txt, "--n_ind", n_ind, "--n_sites", n_sites + 5], ok=False)
        if n_sites > 1:
            run(san_bin, tmp_path, ["--geno", txt, "--n_ind", n_ind, "--n_sites", n_sites - 1], ok=False)
        run(san_bin, tmp_path, ["--geno", txt, "--n_ind", n_ind + 2, "--n_sites", n_sites], ok=False)  # (+1 would take the position column)
        bad = raw.copy()
        bad[n_sites // 2, 0, 1] = np.nan
        nb = str(tmp_path / "nan.bin")
        bad.tofile(nb)
        run(san_bin, tmp_path, ["--geno", nb, "--probs", "--n_ind", n_ind, "--n_sites", n_sites], ok=False)
        case += 1
    # ---- BGZF text: the same lines, inflated block-parallel; small blocks so that lines straddle them
    n_ind, n_sites = 24, 1500
    lines = ["chr\tpos\t" + "\t".join("i%d" % i for i in range(n_ind))]
    for s_ in range(n_sites):
        lines.append("" if s_ == 7 else "c\t%d\t" % s_ + "\t".join(str(int(x)) for x in rng.integers(-1, 3, size=n_ind)))
    data = ("\n".join(lines) + "\n").encode()
    plain = str(tmp_path / "plain.gz")
    with gzip.open(plain, "wb") as fh:
        fh.write(data)
    args = ["--n_ind", n_ind, "--n_sites", n_sites, "--n_boot_rep", 2, "--boot_block_size", 10, "--seed", 3]
    want = run(san_bin, tmp_path, ["--geno", plain] + args)
    want2 = run(san_bin, tmp_path, ["--geno", plain, "--n_gpus", 2] + args)  # (the stub's numbers add up differently)
    for block in (0xff00, 1000, 37):
        bz = str(tmp_path / ("b%d.gz" % block))
        write_bgzf(bz, data, block)
        assert gzip.open(bz, "rb").read() == data  # (a valid multi-member gzip for everybody else)
        for thr in (1, 7):
            assert run(san_bin, tmp_path, ["--geno", bz, "--n_threads", thr] + args) == want
            assert run(san_bin, tmp_path, ["--geno", bz, "--n_threads", thr, "--n_gpus", 2] + args) == want2
        run(san_bin, tmp_path, ["--geno", bz, "--n_ind", n_ind, "--n_sites", n_sites + 3], ok=False)  # premature EOF
        run(san_bin, tmp_path, ["--geno", bz, "--n_ind", n_ind, "--n_sites", n_sites - 3], ok=False)  # not at EOF
    raw_b = bytearray(open(str(tmp_path / "b1000.gz"), "rb").read())
    hurt = str(tmp_path / "hurt.gz")
    raw_b[len(raw_b) // 2] ^= 0x55  # a damaged block in the middle
    open(hurt, "wb").write(bytes(raw_b))
    assert "GENO file" in run(san_bin, tmp_path, ["--geno", hurt, "--n_threads", 4] + args, ok=False)
    cut = str(tmp_path / "cut.gz")
    open(cut, "wb").write(open(str(tmp_path / "b1000.gz"), "rb").read()[:5000])  # the file ends inside a block
    run(san_bin, tmp_path, ["--geno", cut, "--n_threads", 4] + args, ok=False)
    run(san_bin, tmp_path, ["--geno", "/nonexistent", "--n_ind", 3, "--n_sites", 3], ok=False)
    run(san_bin, tmp_path, ["--n_ind", 3, "--n_sites", 3], ok=False)


def test_the_host_says_when_nearly_identical_pairs_were_left_alone(san_bin, tmp_path):
    """ngd_last_fixup() after every engine call (one-image engines: include/ngsdist_amd.h): pairs the fix-up pass had to leave
    alone are a WARNING on stderr, once per run; --verbose 2 also says what was recomputed.  The stub engine reports 5 pairs
    left and 3 recomputed for a data set of 77 individuals, nothing for any other."""
    rng = np.random.default_rng(3)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for n_ind, verbose, want_warn, want_info in ((77, 1, 1, 0), (77, 2, 1, 4), (24, 2, 0, 0)):
        n_sites = 40
        path = str(tmp_path / ("gl_%d.bin" % n_ind))
        rng.random((n_sites, n_ind, 3)).astype(np.float64).tofile(path)
        out = str(tmp_path / "o.dist")
        r = subprocess.run([san_bin, "--geno", path, "--probs", "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--indep_geno",
                            "--n_boot_rep", "3", "--boot_block_size", "4", "--out", out, "--verbose", str(verbose)],
                           capture_output=True, env=env, timeout=300)
        err = r.stderr.decode(errors="replace")
        assert r.returncode == 0 and "Sanitizer" not in err and "runtime error" not in err, err[-2000:]
        assert err.count("WARNING: 5 pairs of nearly identical individuals") == want_warn
        assert (err.count("3 pairs of nearly identical individuals recomputed") >= 1) == (want_info > 0)
        assert n_rows(open(out).read()) == 4 * n_ind
