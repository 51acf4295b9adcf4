import os, sys, numpy as np, torch, scipy.stats as st
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from torchebm_amd import _lib
dev = torch.device("cuda")
for batch, n_noise in ((2048, 205), (1000, 50), (65536, 3276)):
    buf = torch.zeros(batch, 2, device=dev); out = torch.empty(batch, 2, device=dev)
    ps, zs = [], []
    for seed in range(12):
        steps = 300 if batch < 10000 else 60
        hits = np.zeros(batch); adj = 0
        for s in range(steps):
            _lib.call("ebm_pcd_start_points_f32", buf.data_ptr(), batch, 2, out.data_ptr(), batch, 1, n_noise, 0.01, 1000 + seed, 3 * s, None, _lib.stream_handle(dev))
            sel = (out != 0).any(dim=1).cpu().numpy()
            assert sel.sum() == n_noise
            hits += sel; adj += (sel[:-1] & sel[1:]).sum()
        p = n_noise / batch
        chi2 = ((hits - steps * p) ** 2 / (steps * p * (1 - p))).sum()
        ps.append(st.chi2.cdf(chi2, batch - 1))
        pp = n_noise * (n_noise - 1) / (batch * (batch - 1)); npairs = (batch - 1) * steps
        zs.append((adj - npairs * pp) / np.sqrt(npairs * pp * (1 - pp)))
    print(batch, "chi2 cdf per seed:", np.round(ps, 3), " adjacent-pair z:", np.round(zs, 2))
