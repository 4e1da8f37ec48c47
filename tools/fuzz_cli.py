#!/usr/bin/env python3
"""tools/fuzz_cli.py [first_seed] [n_cases] -- random input files and command lines through the C++ host
(ngsdist_amd/bin/ngsDist) against the oracle's restatement of the reference's main loop (oracle.run_reference_flow):
binary / gz-binary / text / gz-text / stdin input, likelihoods or called genotypes, every flag of the reference's command
line that reaches the hot path, every kernel, --prep host|device, --n_gpus N --same_device, --max_device_bytes.
Called-genotype data must print the SAME BYTES (every term is dyadic); likelihood data the same bytes or, where a sum
differs in its last bits, cells within 2e-10.  Test infrastructure; needs a GPU.  Exit 1 if any case is off."""
import gzip
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ngsdist_amd", "bin", "ngsDist")
first, n_cases = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)


def cells(text):
    out = []
    for line in text.split("\n"):
        f = line.split("\t")
        if len(f) > 2:
            out.append([float(x) for x in f[1:]])
    return np.array(out)


bad = same_bytes = 0
t_start = time.time()
tmp = tempfile.mkdtemp(prefix="ngd_fuzz_")
for case in range(first, first + n_cases):
    rng = np.random.default_rng(case)
    n_ind = int(rng.choice([2, 3, 6, 17, 24, 40, 65, 130]))
    n_sites = int(rng.choice([1, 2, 5, 16, 17, 100, 333, 1000, 2500]))
    probs = bool(rng.integers(0, 2))
    huge = rng.integers(0, 15) == 0  # now and then: more lines than one reader group / more bytes than one staging buffer
    text = bool(rng.integers(0, 2)) or not probs  # called genotypes come as text only
    gz = text  # the reference's rule (ngsDist.cpp:82-95): a .gz file is text, anything else is binary likelihoods
    if huge and text:
        n_ind, n_sites = min(n_ind, 24), int(rng.choice([5000, 20000, 40000]))
    elif huge:
        n_ind, n_sites = 130, int(rng.choice([90000, 120000]))  # 280 / 374 MB of doubles
    log_scale = probs and bool(rng.integers(0, 4) == 0)
    called_in = not probs
    miss = float(rng.choice([0.0, 0.05, 0.4]))
    # --- the input file
    if probs:
        p = O.synth_indmajor(5000 + case, n_ind, n_sites, miss_frac=miss)  # [ind][site][3], normalised
        raw = p.transpose(1, 0, 2).copy()  # site-major as in the file
        if log_scale:
            with np.errstate(divide="ignore"):
                raw = np.log(raw)
    else:
        g = rng.integers(0, 3, size=(n_sites, n_ind))
        g[rng.random((n_sites, n_ind)) < miss] = -1
    path = os.path.join(tmp, "in_%d" % case + (".txt" if text else ".bin") + (".gz" if gz else ""))
    opener = (lambda q, m: gzip.open(q, m)) if gz else (lambda q, m: open(q, m))
    n_prefix = int(rng.integers(0, 3)) if text else 0
    header = text and n_prefix > 0 and bool(rng.integers(0, 3) == 0)
    bgzf = text and bool(rng.integers(0, 2))  # blocked gzip (bgzip / htslib): inflated block-parallel by the host
    if text and bgzf:
        import io
        import struct
        import zlib
        sio = io.StringIO()
        if header:
            sio.write("chr\tpos\t" + "\t".join("ind%d" % i for i in range(n_ind * (3 if probs else 1))) + "\n")
        for s in range(n_sites):
            pre = "".join("chr%d\t" % (s % 7) if k == 0 else "pos_%d\t" % s for k in range(n_prefix))
            sio.write(pre + ("\t".join(repr(float(x)) for x in raw[s].reshape(-1)) if probs else "\t".join(str(int(x)) for x in g[s])) + "\n")
        data = sio.getvalue().encode()
        blk = int(rng.choice([0xff00, 4096, 61]))
        with open(path, "wb") as fh:
            for k in list(range(0, len(data), blk)) + [None]:
                chunk = b"" if k is None else data[k:k + blk]
                c = zlib.compressobj(6, zlib.DEFLATED, -15)
                comp = c.compress(chunk) + c.flush()
                fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                         + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    elif text:
        with opener(path, "wt") as fh:
            if header:  # a header is recognised by its non-numeric fields (read_data.cpp:62-70)
                fh.write("chr\tpos\t" + "\t".join("ind%d" % i for i in range(n_ind * (3 if probs else 1))) + "\n")
            for s in range(n_sites):
                pre = "".join("chr%d\t" % (s % 7) if k == 0 else "pos_%d\t" % s for k in range(n_prefix))
                if probs:
                    fh.write(pre + "\t".join(repr(float(x)) for x in raw[s].reshape(-1)) + "\n")
                else:
                    fh.write(pre + "\t".join(str(int(x)) for x in g[s]) + "\n")
    else:
        with opener(path, "wb") as fh:
            fh.write(raw.astype(np.float64).tobytes())
    # --- flags
    call = probs and bool(rng.integers(0, 3) == 0)
    N_thresh, call_thresh = (float(rng.choice([0.0, 0.3, 0.5])), float(rng.choice([0.6, 0.9]))) if call else (0.0, 0.0)
    pdel = bool(rng.integers(0, 2))
    avg = bool(rng.integers(0, 2))
    model = int(rng.integers(0, 3))
    indep_flag = bool(rng.integers(0, 2)) or bool(huge)  # (the oracle's EM is too slow for the large cases)
    indep = indep_flag or call or not probs  # ngsDist.cpp:55-65
    tot = 0 if pdel else int(rng.choice([0, 0, 12345]))  # (the two together are an argument error)
    n_boot = int(rng.choice([0, 0, 1, 3, 35])) if not huge else int(rng.choice([0, 2]))
    B = min(int(rng.choice([1, 1, 4, 10, 16])), n_sites)
    seed = int(rng.integers(1, 1 << 30))
    kernel = str(rng.choice(["stream", "mfma"] if indep else ["em_table", "em_fast", "em_faithful"]))
    args = ["--geno", path, "--n_ind", n_ind, "--n_sites", n_sites, "--seed", seed, "--evol_model", model,
            "--n_threads", int(rng.choice([1, 3, 16])), "--kernel", kernel]
    args += ["--probs"] if probs else []
    args += ["--log_scale"] if log_scale else []
    args += ["--call_geno", "--N_thresh", N_thresh, "--call_thresh", call_thresh] if call else []
    args += ["--pairwise_del"] if pdel else []
    args += ["--avg_nuc_dist"] if avg else []
    args += ["--indep_geno"] if indep_flag else []
    args += ["--tot_sites", tot] if tot else []
    args += ["--n_boot_rep", n_boot, "--boot_block_size", B] if n_boot else []
    how = int(rng.integers(0, 5))
    if how == 1:
        args += ["--n_gpus", int(rng.choice([2, 3])), "--same_device"]
    elif how == 2:
        n_pad = (n_ind + 127) // 128 * 128
        args += ["--max_device_bytes", (512 << 20) + 256 * n_pad * n_pad * 8 + 64 * n_ind * n_ind + 400 * max(n_ind, 24) * max(400, n_sites // 3)]
    elif how == 3:
        args += ["--prep", str(rng.choice(["host", "device"]))]
    # (round 5, a generator of its own: earlier cases keep their command lines) the MFMA engine on one operand image in
    # congruent coordinates + the fix-up pass of nearly identical pairs, or on two images, whatever the engine would pick
    rng_r5 = np.random.default_rng(5_000_000 + case)
    if kernel == "mfma":
        args += [[], ["--single_image"], ["--two_images"], []][int(rng_r5.integers(0, 4))]
    labels = None
    lab = int(rng.integers(0, 4))
    if lab:
        labels = ["s%d_%s" % (i, "ab*+#"[i % 5]) for i in range(n_ind)]
        lpath = os.path.join(tmp, "labels_%d.txt" % case)
        with open(lpath, "w") as fh:
            if lab == 2:
                fh.write("a header line\n")
            for i, l in enumerate(labels):
                fh.write(l + ("\tignored column" if i % 3 == 0 else "") + "\n")
        args += ["--labelsH" if lab == 2 else "--labels", lpath]
    use_stdin = (not text) and (not gz) and bool(rng.integers(0, 4) == 0)
    tag = (case, n_ind, n_sites, "text" if text else "bin", "gz" if gz else "", [str(a) for a in args[10:]])
    # --- expected text
    try:
        kw = dict(in_logscale=log_scale, call_geno=call, N_thresh=N_thresh, call_thresh=call_thresh)
        if text:
            pp = O.load_text(path, n_ind, n_sites, probs, **kw)
        else:
            pp = O.prep_binary(raw, n_ind, n_sites, **kw)
        exp = O.run_reference_flow(pp, labels=labels, score=O.score_matrix(avg), pairwise_del=pdel, indep_geno=indep, tot_sites=tot,
                                   evol_model=model, n_boot_rep=n_boot, boot_block_size=B, seed=seed, n_threads=8)
        out = os.path.join(tmp, "out_%d.dist" % case)
        a = [str(x) for x in args]
        if use_stdin:
            a[1] = "-"
        r = subprocess.run([BIN] + a + ["--out", out, "--verbose", "0"], capture_output=True,
                           stdin=open(path, "rb") if use_stdin else None, timeout=300)
        if r.returncode != 0:
            bad += 1
            print("EXIT", r.returncode, tag, r.stderr.decode()[-300:], flush=True)
            continue
        got = open(out).read()
        os.remove(out)
        if got == exp:
            same_bytes += 1
        else:
            dyadic = called_in or (call and call_thresh == 0.0)
            ca, cb = cells(got), cells(exp)
            # cells are compared BEFORE the logarithm of the evolutionary model: a pair at saturation (d = 1, or 3/4 under
            # JC69) prints inf, nan or 36.7 / 27.6 depending on the last bit of its sum, here as in the reference
            back = (lambda x: x) if model == 0 else (lambda x: np.exp(-x)) if model == 1 else (lambda x: np.exp(-x * 4 / 3))
            with np.errstate(all="ignore"):
                sat = (model > 0) & ((~np.isfinite(ca)) | (ca > 25)) & ((~np.isfinite(cb)) | (cb > 25))  # both at saturation
                close = ca.shape == cb.shape and bool(np.all((np.abs(ca - cb) <= 2e-10) | (ca == cb) | (np.isnan(ca) & np.isnan(cb)) |
                                                             (np.abs(back(ca) - back(cb)) <= 1e-9) | sat))
            # (missing x missing sites without --pairwise_del put non-dyadic thirds into called-genotype sums too)
            if not close or (dyadic and pdel):
                bad += 1
                print("DIFF" if close else "MISMATCH", tag, flush=True)
                if ca.shape != cb.shape:
                    print("   shapes", ca.shape, cb.shape, "lines", got.count("\n"), exp.count("\n"), flush=True)
                else:
                    with np.errstate(all="ignore"):
                        okm = (np.abs(ca - cb) <= 2e-10) | (ca == cb) | (np.isnan(ca) & np.isnan(cb)) | (np.abs(back(ca) - back(cb)) <= 1e-9) | sat
                    off = np.argwhere(~okm)
                    print("   %d cells off, first:" % len(off), [(tuple(int(x) for x in k), float(ca[tuple(k)]), float(cb[tuple(k)])) for k in off[:4]], flush=True)
                if os.environ.get("FUZZ_KEEP"):
                    open(os.path.join(tmp, "got_%d.dist" % case), "w").write(got)
                    open(os.path.join(tmp, "exp_%d.dist" % case), "w").write(exp)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(ex), flush=True)
    finally:
        if os.path.exists(path):
            os.remove(path)
    if (case - first) % 50 == 49:
        print("... %d cases, %d same bytes, %d bad, %.0f s" % (case - first + 1, same_bytes, bad, time.time() - t_start), flush=True)
print("fuzz_cli: %d cases from seed %d: %d printed the same bytes, %d bad" % (n_cases, first, same_bytes, bad))
sys.exit(1 if bad else 0)
