"""The block-box arithmetic behind the forward tile kernel's dense rounds (sls_common.hpp: make_block_box,
sls_sort.hip: block_mask_of), restated in NumPy and checked against the test it has to be a superset of: the tile
kernels' cull_pass on a whole 8x2 pixel block (sls_tile.hpp).  CPU only — the device code itself is exercised end to end
by the engine-vs-checker tests with block_masks forced on (tests/test_timed_path.py)."""
import numpy as np
import pytest


def make_block_box(cx, cy, ex, ey, NC):
    if not (ex >= 0.0 and ey >= 0.0):
        return None
    f = np.float32
    lo = np.ceil((f(cx) - f(ex) - f(7.01)) * f(0.125))
    hi = np.floor((f(cx) + f(ex) + f(0.01)) * f(0.125))
    n = int(min(hi - lo, NC - 1))
    c0 = int(max(min(lo, 1.0e6), -1.0e6)) % NC
    r0 = int(min(max(np.ceil((f(cy) - f(ey) - f(1.01)) * f(0.5)), 0.0), 63.0))
    r1 = int(min(np.floor((f(cy) + f(ey) + f(0.01)) * f(0.5)), 63.0))
    if n < 0 or r1 < r0:
        return None
    return c0, n, r0, r1


def block_mask_of(box, tx, ty, NC):
    if box is None:
        return 0
    c0, n, r0, r1 = box
    a, b = max(r0 - ty * 8, 0), min(r1 - ty * 8, 7)
    ym = (((4 << (2 * b)) - 1) & ~((1 << (2 * a)) - 1)) if a <= b else 0
    d0 = (2 * tx - c0) % NC
    d1 = (d0 + 1) % NC
    xm = (0x5555 if d0 <= n else 0) | (0xAAAA if d1 <= n else 0)
    return ym & xm


def cull_pass_block(cx, cy, ex, ey, x0, y0, W, wrap):
    """cull_pass (sls_tile.hpp) against the full 8x2 block at (x0, y0): centre (x0 + 3.5, y0 + 0.5), half extents (3.5, 0.5)."""
    f = np.float32
    dx0 = f(x0) + f(3.5) - f(cx)
    dxc = dx0 - (f(W) * np.rint(dx0 / f(W)) if wrap else f(0.0))
    return abs(dxc) <= f(ex) + f(3.5) and abs(f(y0) + f(0.5) - f(cy)) <= f(ey) + f(0.5)


@pytest.mark.parametrize("W,H,wrap", [(1024, 64, True), (2048, 64, True), (512, 128, True), (208, 48, False)])
def test_block_masks_cover_cull_pass(W, H, wrap):
    rng = np.random.default_rng(W + H)
    GX, GY = (W + 15) // 16, (H + 15) // 16
    NC = GX * 2
    missed = extra = total = 0
    for _ in range(600):
        cx, cy = rng.uniform(-2.0, W + 2.0), rng.uniform(-2.0, H + 2.0)
        ex, ey = rng.choice([0.55, 1.3, 3.7, 9.2, 40.0, W * 0.6]), rng.choice([0.55, 1.1, 2.4, 6.0, 30.0])
        if rng.random() < 0.05:
            ex = -1.0e30
        box = make_block_box(cx, cy, ex, ey, NC)
        for ty in range(GY):
            for tx in range(GX):
                m = block_mask_of(box, tx, ty, NC)
                for by in range(8):
                    for bx in range(2):
                        x0, y0 = tx * 16 + 8 * bx, ty * 16 + 2 * by
                        if x0 >= W or y0 >= H:
                            continue
                        want = ex >= 0 and cull_pass_block(cx, cy, ex, ey, x0, y0, W, wrap)
                        got = bool((m >> (2 * by + bx)) & 1)
                        total += 1
                        missed += int(want and not got)
                        extra += int(got and not want)
    assert missed == 0, f"{missed} blocks pass cull_pass but are not in the mask"
    # conservative, not sloppy: what the mask adds lies within the 0.01 px of slack or, on images with edges, is the
    # wrap-around the integer ranges apply there too (a box at the left edge also marks the right edge's blocks)
    assert extra <= (0.002 if wrap else 0.03) * total, (extra, total)
