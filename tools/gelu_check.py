"""Accuracy of the GELU used by the GEMM epilogues (csrc/bc_common.h: Abramowitz-Stegun 7.1.26 erfc) against the fp64 erf form."""
import numpy as np, math
x = np.linspace(-12, 12, 2000001).astype(np.float32)
z = np.abs(x) * np.float32(0.70710678118654752)
t = (1.0 / (1.0 + np.float32(0.3275911) * z)).astype(np.float32)
poly = t * (np.float32(0.254829592) + t * (np.float32(-0.284496736) + t * (np.float32(1.421413741) + t * (np.float32(-1.453152027) + t * np.float32(1.061405429)))))
e = np.exp2(-(z * z) * np.float32(1.4426950408889634)).astype(np.float32)
erfc = (poly * e).astype(np.float32)
g = np.where(x >= 0, 0.5 * x * (2.0 - erfc), 0.5 * x * erfc).astype(np.float32)
from scipy.special import erf
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / math.sqrt(2)))
err = np.abs(g - ref)
print("max abs err", err.max(), "at", x[err.argmax()], "max rel err where |ref|>1e-3:", (err / np.maximum(np.abs(ref), 1e-30))[np.abs(ref) > 1e-3].max())
