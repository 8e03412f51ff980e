/* ORACLE (test infrastructure only — never linked into or called by the product path).
 *
 * Bit-exact CPU restatement of the per-row MLP of the PointNet pool (reference
 * seq_lattice/lattice_modules.py:460-473: Linear -> ReLU -> Linear -> ReLU -> Linear) in the one
 * summation order DESIGN.md §3.8 fixes for it:
 *
 *     acc = bias[o];  for i = 0 .. cin-1:  acc = fma(x[i], W[o][i], acc)      (IEEE fp32 fused multiply-add)
 *
 * Why the order has to be pinned at all: lm:513-525 gathers the barycentric weight of the ARG-max row, so a
 * value that differs in its last bit can flip an arg-max between two near-equal rows and change a whole output
 * element by O(1) (2M maxima per frame make near-ties a daily event).  With the order fixed the HIP kernel
 * (csrc/pool.hip) and this file produce identical bits, and the pooled tensor is compared exactly.
 *
 * Built by oracle/Makefile into oracle/_build/libpool_mlp.so; oracle/ops.py loads it through ctypes and falls
 * back to a (slow) float64 emulation of the same fma chain when the library is missing.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* x [rows, cin] -> y [rows, cout]; W [cout, cin] (torch Linear layout), b [cout]; relu != 0 clamps at 0 */
void oracle_linear_fma(const float* x, int64_t rows, int cin, const float* W, const float* b, int cout, int relu,
                       float* y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    const float* xr = x + (size_t)r * cin;
    float* yr = y + (size_t)r * cout;
    for (int o = 0; o < cout; ++o) {
      const float* w = W + (size_t)o * cin;
      float acc = b ? b[o] : 0.0f;
      for (int i = 0; i < cin; ++i) acc = fmaf(xr[i], w[i], acc);
      yr[o] = (relu && !(acc > 0.0f)) ? 0.0f : acc;
    }
  }
}
