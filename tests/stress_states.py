"""Synthetic SDF-network states that push one layer's activations into the fp16 subnormal range of the f16x3 kernels
(test infrastructure).

The wide kernels hand a layer's activations to the next layer as fp16 pairs hi + lo with the residual lo UNSCALED
(csrc/gen_mlp32.py split_ops): in the scaled domain u = h * 100 / ln 2, lo is an fp16 subnormal whenever |u| < 2^-3 and hi
itself is one below 2^-14.  ``subnormal_stress_state`` biases layer ``layer`` to z ~ -c (softplus_100(-0.05) = 6.7e-5,
softplus_100(-0.11) = 1.7e-7) and multiplies the NEXT layer's gain so that those tiny activations still decide its output:
a kernel (or an MFMA / conversion / AGPR path) that flushed fp16 subnormals would be off by 6e-5 (residuals lost) or 3e-4
(everything lost) in the sdf - the numpy emulation with a flushing split shows exactly that (tests/test_packing32_emulated.py)."""
import numpy as np

# (c, eps, gain):  level -> bias -c on layer L, its weight-norm gain x eps, layer L + 1's gain x `gain`
LEVELS = {
    "residuals_subnormal": (0.05, 0.02, 2000.0),   # u in [2e-3, 5e-2]: hi normal, EVERY residual an fp16 subnormal
    "all_subnormal": (0.11, 0.02, 2000.0),         # u in [5e-6, 1.3e-4]: hi subnormal too, residuals below the smallest subnormal
}


def subnormal_stress_state(state, level: str, layer: int = 5):
    c, eps, gain = LEVELS[level]
    st = {k: np.array(v, copy=True) for k, v in state.items()}
    st[f"sdf_network.lin{layer}.bias"][:] = -c
    st[f"sdf_network.lin{layer}.weight_g"] *= eps
    st[f"sdf_network.lin{layer + 1}.weight_g"] *= gain
    return st
