"""Seeded recipes shared by the SegFormer training parity tests (same draws as tools/make_goldens.py)."""

import numpy as np
import torch


def mit_drop_masks(depths, rate, batch, seed):
    total = sum(depths)
    g = np.random.default_rng([seed, total, batch, 13])
    dpr = torch.linspace(0, rate, total).tolist()
    masks = []
    for i in range(total):
        pair = []
        for j in range(2):
            m = (g.uniform(size=batch) < 1.0 - dpr[i]).astype(np.float32)
            if i == total - 1 and j == 0:
                m[0] = 0.0
            if i == total // 2 and j == 1:
                m[-1] = 0.0
            pair.append(torch.from_numpy(m))
        masks.append(tuple(pair))
    return masks


def chan_mask(batch, ch, seed):
    g = np.random.default_rng([seed, batch, ch, 11])
    return torch.from_numpy((g.uniform(size=(batch, ch)) < 0.9).astype(np.float32))


def grad_sample(g, n):
    f = g.detach().flatten()
    step = max(1, f.numel() // n)
    return f[::step][:n].contiguous().float().cpu().numpy()


def check_grads(named_grads, golden, *, tight=(), tol=2e-2, tight_tol=1e-3, n=512, frac=0.99):
    """Every gradient sample vs the golden: |diff| <= tol * max|ref| on >= ``frac`` of the sampled
    elements, and the L2 norm within ``tol``.  Returns the worst relative error seen."""
    worst = 0.0
    for name, grad in named_grads:
        ref = golden["grad/" + name]
        got = grad_sample(grad, n)
        assert got.shape == ref.shape, name
        t = tight_tol if name.startswith(tuple(tight)) else tol
        scale = max(float(np.abs(ref).max()), 1e-6)   # floor: analytically-zero gradients are rounding noise
        rel = np.abs(got - ref) / scale
        ok = (rel <= t).mean()
        assert ok >= frac, f"{name}: only {ok:.3f} of sampled grads within {t} (max rel {rel.max():.3e})"
        gn, rn = float(grad.double().norm().item()), float(golden["gradnorm/" + name])
        assert abs(gn - rn) <= t * rn + 2e-6, f"{name}: grad norm {gn} vs {rn}"
        worst = max(worst, float(np.quantile(rel, frac)))
    return worst
