"""Host-side logic of the engine that needs no GPU."""


def test_status_parsers_agree():
    """The engine parses the iteration's status block two ways — from a torch tensor (event path, sharded path) and
    from the NumPy view of the pinned mirror (polled path): same dictionary."""
    import numpy as np
    import torch
    from splat_loam_amd.engine import MappingEngine
    rng = np.random.default_rng(0)
    for flags in (0, 1, 2, 4, 3):
        h = np.zeros(8, np.int32)
        h[0] = np.int32(np.uint32(3_000_000_000).astype(np.int32)) if flags == 1 else 123456      # (R beyond 2^31 too)
        h[1] = flags
        h[2:7] = rng.normal(size=5).astype(np.float32).view(np.int32)
        h[7] = 4242
        a = MappingEngine._parse_status(torch.from_numpy(h.copy()))
        b = MappingEngine._parse_status_np(h.copy())
        assert a == b, (a, b)
        assert a["overflow"] == bool(flags) and a["exchange_count"] == 4242
