"""Generate tests/golden/tiny_fact_golden.npz from the CPU oracle (oracle/fact_oracle.py, fp64).

The reference (TF2/Keras/Orbit) cannot run in this environment, so these vectors are produced BY the
oracle: they pin the oracle (and through it the HIP engine) against drift, not against the reference.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fact_oracle as O  # noqa: E402


def main():
    cfg = O.TINY_CFG
    params = O.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k, v in params.items():  # non-trivial biases / LN affine
        if k.endswith("/bias") or k.endswith("/beta"):
            v.copy_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
        elif k.endswith("/gamma"):
            v.copy_(1.0 + torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1)
    batch = O.synthetic_batch(cfg, 2, 8, seed=11)
    loss, grads, pred = O.loss_and_grads(params, cfg, batch["motion_input"], batch["audio_input"], batch["target"])
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(x) for k, x in params.items()}
    p1, _, _ = O.adam_update(params, grads, m, v, 0, 1e-3)
    ar = O.infer_auto_regressive(params, cfg, batch["motion_input"], torch.cat(
        [batch["audio_input"], batch["audio_input"][:, :3]], dim=1), steps=4)
    out = {
        "pred": pred.float().numpy(), "loss": np.float64(loss), "ar": ar.float().numpy(),
        "grad_norms": np.array([float(grads[n].norm()) for n, _ in O.param_shapes(cfg)]),
        "grad_sums": np.array([float(grads[n].sum()) for n, _ in O.param_shapes(cfg)]),
        "adam_delta_norms": np.array([float((p1[n] - params[n]).norm()) for n, _ in O.param_shapes(cfg)]),
        "all_ones_row0": O.fact_forward(params, cfg, torch.ones(1, 32, 225, dtype=torch.float64),
                                        torch.ones(1, 64, 35, dtype=torch.float64))[0, 0].float().numpy(),
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_fact_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; loss", float(loss))


if __name__ == "__main__":
    main()
