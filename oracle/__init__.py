"""oracle/ — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package, and only as the checker / the timed CPU baseline.  The product package
(`upflow_pytorch_amd/`) never imports it and has no CPU fallback: it raises when the HIP library is
missing.

Parity status: PINNED.  The reference itself holds no tests or golden vectors for this path
(SURVEY.md §4), and its native CUDA path cannot be compiled here (needs nvcc,
`model/correlation_package/setup.py:20-25`), so the pins are outputs of the reference's own Python
path imported in the build container: `tests/golden/make_golden.py` (committed) imports
`/root/reference` and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every
function here against those vectors (correlation and warp-mask bit-exact, the rest <= 1e-6).

Third-party arithmetic on the path that is not under /root/reference: ATen `grid_sample`,
`interpolate`, `unfold` (torch==1.1.0 pinned by `requirements.txt:12`; torch 2.10 binaries here).
Their published algorithms are restated explicitly in `oracle/ops.py` (no call to grid_sample or
interpolate), anchored on the reference's call sites `model/pwc_modules.py:74,79,101,200,205`,
`utils/tools.py:1257,1261,1304`, `utils/pytorch_correlation.py:30-31,38`.
"""
from .ops import (corr81, corr81_unfold, corr81_backward, correlation_general, correlation_well_defined, correlation_forward_literal,
                  correlation_backward_supported, correlation_general_backward, correlation_backward_literal, warp, warp_backward,
                  flow_upsample, sgu_blend, normalize_pair, occ_check, census_distance, epe,
                  boundary_warp, robust_loss_sums, smooth_edge1)  # noqa: F401
