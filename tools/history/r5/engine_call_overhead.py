import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
import ngsdist_amd as N
dev = torch.device("cuda", 0)
def run(name, n_ind, n_sites, indep, n_boot, block, kernel):
    n_pairs = N.n_pairs(n_ind)
    eng = N.Engine(n_ind, n_sites, indep_geno=indep, kernel=kernel); eng.synth_fill(3)
    n_mat = n_boot + 1
    d_all = torch.zeros((n_mat, n_pairs), dtype=torch.float64, device=dev); d_c = torch.zeros((n_mat, n_pairs), dtype=torch.int64, device=dev)
    rng = N.Taus(12345)
    if n_boot:
        nb = n_sites // block
        mult = np.stack([np.ones(nb, dtype=np.uint32)] + [np.bincount(rng.block_map(nb).astype(np.int64), minlength=nb).astype(np.uint32) for _ in range(n_boot)])
    walls, devs = [], []
    for it in range(8):
        eng.drop_caches()
        torch.cuda.synchronize()
        t = time.perf_counter()
        if n_boot: eng.run_batch(mult=mult, block_size=block, d_sum_ptr=d_all.data_ptr(), d_cnt_ptr=d_c.data_ptr())
        else: eng.run_device(d_all.data_ptr(), d_c.data_ptr())
        w = time.perf_counter() - t
        tm = eng.timing()
        walls.append(w * 1e3); devs.append(tm["ms_total"])
    print("%-6s engine call wall %.3f ms, device (events) %.3f ms, accum %.3f reduce %.3f count %.3f -> host overhead %.3f ms" % (
        name, np.median(walls[2:]), np.median(devs[2:]), tm["ms_accum"], tm["ms_reduce"], tm["ms_count"], np.median(walls[2:]) - np.median(devs[2:])))
    eng.close()
run("cfg2", 200, 100000, True, 0, 1, "mfma")
run("cfg5", 500, 500000, True, 64, 1000, "mfma")
run("cfg3/8", 1000, 125000, True, 0, 1, "mfma")
run("cfg3", 1000, 1000000, True, 0, 1, "mfma")
