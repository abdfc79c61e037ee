"""BASELINE.json configs[1] and configs[3] against the fp64 oracle, in composition: the same models the
bench runs — d256 / 2+2 / kernel_sizes [11,5,3] / T384 / F224 and d512 / 6+6 / 8 heads / T512 / F224 — at B=2, where the
oracle's training step takes seconds.  These shapes take the kernels written for them (A-stationary K=256/512/1024 GEMMs,
one-pass attention backward at T=384 / dh=32 and T=512 / dh=64, `ctc_kernel` with three state registers), which the small
configurations of test_model_gpu.py reach only through test_ops_gpu.py.  f32 to the north_star's fp32 tolerance (1e-4 on
the logits), bf16 to the tolerance stated in test_model_gpu.BF16_TOL / CFG4_BF16_TOL.  PARITY UNPINNED against TensorFlow
(not installable)."""
import pytest

from test_model_gpu import check_train_step, BF16_TOL

pytestmark = pytest.mark.gpu

CFG2 = dict(dim=256, num_conv_squeeze_blocks=2, num_conv_conform_blocks=2, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
            num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(384, 224), B=2)
CFG4 = dict(dim=512, num_conv_squeeze_blocks=6, num_conv_conform_blocks=6, kernel_sizes=[11, 5, 3], num_conv_per_block=3,
            num_heads=8, expansion_factor=2, transformer_kernel_size=15, input_shape=(512, 224), B=2)
# 48 modules deep (36 Conv1DBlocks + 12 transformer blocks), ~250 chained bf16 activations
CFG4_BF16_TOL = dict(BF16_TOL, logits=0.3, grad=0.2, grad_small=0.35, loss=5e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_config2_train_step_vs_oracle(dtype, dropout):
    check_train_step(CFG2, dtype, dropout, "config2_B2", check_decode=True)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_config4_train_step_vs_oracle(dtype):
    check_train_step(CFG4, dtype, 0.2, "config4_B2", bf16_tol=CFG4_BF16_TOL, check_decode=True)


def test_config2_b64_train_step_vs_oracle():
    """configs[1]'s model at B = 64 against the fp64 oracle: the batch at which the project conv's weight-gradient GEMM (gemm.hip TnPsa) folds TWO
    samples per M-split — sample boundaries inside the kernel's step loop, parked accumulators, the per-sample statistics of dh4 — and the
    forward GEMM prologues run with split columns.  (B = 2 above exercises one sample per split; tests/test_full_size_gpu.py compares the two
    HIP routes with each other at this batch.)  bf16, dropout on; ~20 s of oracle time."""
    check_train_step(dict(CFG2, B=64), "bf16", 0.2, "config2_B64")
