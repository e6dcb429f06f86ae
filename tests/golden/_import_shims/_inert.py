class Inert:
    """Importable name that refuses to be used."""
    def __init__(self, name):
        self._name = name
    def __call__(self, *a, **k):
        raise RuntimeError(f"{self._name} is an inert import shim (off the hot path)")
    def __getattr__(self, item):
        return Inert(f"{self._name}.{item}")
