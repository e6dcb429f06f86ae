"""numba decorators as pass-throughs (only decorates dataset-side helpers, never run here)."""
def _passthrough(*dargs, **dkw):
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]
    return lambda f: f
jit = njit = _passthrough
prange = range
class _T:
    def __getattr__(self, n): return self
    def __call__(self, *a, **k): return self
    def __getitem__(self, k): return self
float32 = float64 = int32 = int64 = boolean = void = _T()
