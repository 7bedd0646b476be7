"""ctypes front-end of the CPU checker (oracle/sls_oracle.c).

TEST INFRASTRUCTURE — see the header of sls_oracle.c.  Imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  "parity
unpinned": the reference rasterizer's source is not vendored in
/root/reference.

All arrays are NumPy, C-contiguous.  `Oracle(np.float32)` is the parity
checker for the HIP kernels, `Oracle(np.float64)` verifies analytic gradients.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

REC_STRIDE = 20
GREC_STRIDE = 16
NEAR = float(np.float32(0.2))  # SLS_NEAR / SLS_FAR are float literals
FAR = 100.0
TILE_CULL_MIN_DEFAULT = 0   # SLS_TILE_CULL_MIN_DEFAULT (include/sls_spec.h): 0 = the test is off by default


def build(force: bool = False) -> None:
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (make -C oracle)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, f)) for f in ("liboracle_f32.so", "liboracle_f64.so"))
    if not need:
        src = max(os.path.getmtime(os.path.join(_HERE, "sls_oracle.c")),
                  os.path.getmtime(os.path.join(_HERE, "..", "include", "sls_det_math.h")),
                  os.path.getmtime(os.path.join(_HERE, "..", "include", "sls_spec.h")))
        need = any(os.path.getmtime(os.path.join(_HERE, f)) < src
                   for f in ("liboracle_f32.so", "liboracle_f64.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def should_wrap(fx: float, W: int, tile_w: int) -> bool:
    """D5: azimuth wraps when the image spans 360 deg (|fx|*2pi == W within a
    pixel) and the tile grid divides the width."""
    return abs(abs(fx) * 2.0 * math.pi - W) <= 1.0 and W % tile_w == 0


class Camera:
    """icam/fcam arrays of sls_oracle.c from the reference's matrices.

    viewmatrix = inv(world_T_lidar)^T and projmatrix[:3,:3] = K^T
    (scene/cameras.py:43-50): p_view = R_vw p + t with R_vw = viewmatrix[:3,:3]^T,
    t = viewmatrix[3,:3]; K = projmatrix[:3,:3]^T.
    """

    def __init__(self, H, W, viewmatrix, projmatrix, scale_modifier=1.0, tile=(16, 16),
                 wrap=None, dtype=np.float32, pix_offset=(0.0, 0.0), tile_cull=True):
        V = np.asarray(viewmatrix, dtype=np.float64)
        Pm = np.asarray(projmatrix, dtype=np.float64)
        K = Pm[:3, :3].T
        assert abs(K[0, 1]) < 1e-12 and abs(K[1, 0]) < 1e-12, "skewed K unsupported"
        self.H, self.W = int(H), int(W)
        self.tile = (int(tile[0]), int(tile[1]))
        self.fx, self.fy, self.cx, self.cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
        # D1 as a parameter (SlsCamera.pix_offset): pixel (c, r) sits at image coordinate (c + ox, r + oy) — the
        # rasterizer works with the principal point (cx - ox, cy - oy), ONE subtraction in the working precision
        self.pix_offset = (float(pix_offset[0]), float(pix_offset[1]))
        dt = np.dtype(dtype).type
        self.cx = float(dt(self.cx) - dt(self.pix_offset[0]))
        self.cy = float(dt(self.cy) - dt(self.pix_offset[1]))
        if wrap is None:
            wrap = should_wrap(self.fx, self.W, self.tile[0])
        self.wrap = int(bool(wrap))
        self.dtype = np.dtype(dtype)
        # D10: the binning drops instances whose tile the footprint cannot reach, for rectangles of at least this many
        # tiles (True: SLS_TILE_CULL_MIN_DEFAULT of include/sls_spec.h; False / 0: off)
        self.tile_cull = TILE_CULL_MIN_DEFAULT if tile_cull is True else int(tile_cull or 0)
        self.icam = np.array([self.H, self.W, self.tile[0], self.tile[1], self.wrap, self.tile_cull], dtype=np.int32)
        Rvw = V[:3, :3].T
        tvw = V[3, :3]
        self.fcam = np.concatenate([
            [self.fx, self.fy, self.cx, self.cy, scale_modifier, NEAR, FAR], Rvw.reshape(-1), tvw
        ]).astype(self.dtype)
        self.GX = (self.W + self.tile[0] - 1) // self.tile[0]
        self.GY = (self.H + self.tile[1] - 1) // self.tile[1]
        self.T = self.GX * self.GY


class Oracle:
    def __init__(self, dtype=np.float32, threads: int | None = None):
        build()
        self.dtype = np.dtype(dtype)
        name = "liboracle_f32.so" if self.dtype == np.float32 else "liboracle_f64.so"
        self.lib = C.CDLL(os.path.join(_HERE, name))
        self.lib.or_count_instances.restype = C.c_uint64
        self.lib.or_max_threads.restype = C.c_int
        assert self.lib.or_real_bytes() == self.dtype.itemsize
        if threads is not None:
            self.lib.or_set_threads(int(threads))

    # ------------------------------------------------------------------
    def set_threads(self, n: int) -> None:
        self.lib.or_set_threads(int(n))

    def max_threads(self) -> int:
        return int(self.lib.or_max_threads())

    def _r(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    def camera(self, H, W, viewmatrix, projmatrix, scale_modifier=1.0, tile=(16, 16), wrap=None, pix_offset=(0.0, 0.0),
               tile_cull=True):
        return Camera(H, W, viewmatrix, projmatrix, scale_modifier, tile, wrap, self.dtype, pix_offset, tile_cull)

    def ray_tables(self, cam: Camera):
        col = np.empty((cam.W, 2), self.dtype)
        row = np.empty((cam.H, 2), self.dtype)
        self.lib.or_ray_tables(_ptr(cam.icam), _ptr(cam.fcam), _ptr(col), _ptr(row))
        return col, row

    def preprocess(self, cam: Camera, means, scales, rots, opac, tables=None):
        means, scales, rots, opac = map(self._r, (means, scales, rots, opac))
        N = means.shape[0]
        col, row = tables if tables is not None else self.ray_tables(cam)      # (D10 takes its tile directions from them)
        out = dict(
            rec=np.empty((N, REC_STRIDE), self.dtype), radii=np.empty(N, np.int32),
            rect=np.empty((N, 4), np.int32), tiles=np.empty(N, np.uint32), tmask=np.empty(N, np.uint64),
            depth=np.empty(N, self.dtype))
        self.lib.or_preprocess(_ptr(cam.icam), _ptr(cam.fcam), C.c_int(N), _ptr(means), _ptr(scales),
                               _ptr(rots), _ptr(opac.reshape(-1)), _ptr(col), _ptr(row), _ptr(out["rec"]),
                               _ptr(out["radii"]), _ptr(out["rect"]), _ptr(out["tiles"]), _ptr(out["tmask"]),
                               _ptr(out["depth"]))
        return out

    def bin_sort(self, cam: Camera, pre):
        N = pre["tiles"].shape[0]
        R = int(self.lib.or_count_instances(C.c_int(N), _ptr(pre["tiles"])))
        depth_bits = np.ascontiguousarray(pre["depth"].astype(np.float32)).view(np.uint32)
        out = dict(R=R, keys_unsorted=np.empty(R, np.uint64), vals_unsorted=np.empty(R, np.uint32),
                   keys=np.empty(R, np.uint64), vals=np.empty(R, np.uint32),
                   ranges=np.zeros((cam.T, 2), np.uint32))
        self.lib.or_emit_sort(_ptr(cam.icam), C.c_int(N), _ptr(pre["rect"]), _ptr(pre["tiles"]), _ptr(pre["tmask"]),
                              _ptr(depth_bits), C.c_uint64(R), _ptr(out["keys_unsorted"]),
                              _ptr(out["vals_unsorted"]), _ptr(out["keys"]), _ptr(out["vals"]),
                              _ptr(out["ranges"]))
        return out

    def render_fwd(self, cam: Camera, tables, binned, rec, frag_tol=1e-4):
        col, row = tables
        P = cam.H * cam.W
        out = dict(allmap=np.zeros((7, cam.H, cam.W), self.dtype), pixT=np.ones(P, self.dtype),
                   pixN=np.zeros(P, np.uint32), pixMed=np.zeros(P, np.uint32),
                   pixM1=np.zeros(P, self.dtype), pixM2=np.zeros(P, self.dtype),
                   fragile=np.zeros(P, np.uint8), tile_consumed=np.zeros(cam.T, np.uint32))
        self.lib.or_render_fwd(_ptr(cam.icam), _ptr(cam.fcam), _ptr(col), _ptr(row), _ptr(binned["ranges"]),
                               _ptr(binned["vals"]), _ptr(rec), _ptr(out["allmap"]), _ptr(out["pixT"]),
                               _ptr(out["pixN"]), _ptr(out["pixMed"]), _ptr(out["pixM1"]), _ptr(out["pixM2"]),
                               _ptr(out["fragile"]), _ptr(out["tile_consumed"]), C.c_double(frag_tol))
        out["fragile"] = out["fragile"].reshape(cam.H, cam.W).astype(bool)
        return out

    def render_bwd(self, cam: Camera, tables, binned, rec, fwd, dL_dallmap, want_abs=True, threads=1):
        col, row = tables
        N = rec.shape[0]
        dL = self._r(dL_dallmap)
        grec = np.zeros((N, GREC_STRIDE), self.dtype)
        gabs = np.zeros((N, GREC_STRIDE), self.dtype) if want_abs else None
        self.lib.or_render_bwd(_ptr(cam.icam), _ptr(cam.fcam), _ptr(col), _ptr(row), _ptr(binned["ranges"]),
                               _ptr(binned["vals"]), _ptr(rec), _ptr(fwd["pixT"]), _ptr(fwd["pixN"]),
                               _ptr(fwd["pixMed"]), _ptr(fwd["pixM1"]), _ptr(fwd["pixM2"]), _ptr(dL),
                               C.c_int(N), _ptr(grec), _ptr(gabs), C.c_int(threads))
        return grec, gabs

    def preprocess_bwd(self, cam: Camera, means, scales, rots, radii, grec):
        means, scales, rots, grec = map(self._r, (means, scales, rots, grec))
        N = means.shape[0]
        dm = np.empty((N, 3), self.dtype)
        ds = np.empty((N, 2), self.dtype)
        dr = np.empty((N, 4), self.dtype)
        do = np.empty((N, 1), self.dtype)
        self.lib.or_preprocess_bwd(_ptr(cam.icam), _ptr(cam.fcam), C.c_int(N), _ptr(means), _ptr(scales),
                                   _ptr(rots), _ptr(np.ascontiguousarray(radii, np.int32)), _ptr(grec),
                                   _ptr(dm), _ptr(ds), _ptr(dr), _ptr(do))
        return dm, ds, dr, do

    # ------------------------------------------------------------------
    def forward(self, cam: Camera, means, scales, rots, opac, frag_tol=1e-4):
        """Whole forward: returns a state dict with radii, allmap and everything
        the backward needs."""
        tables = self.ray_tables(cam)
        pre = self.preprocess(cam, means, scales, rots, opac, tables)
        binned = self.bin_sort(cam, pre)
        fwd = self.render_fwd(cam, tables, binned, pre["rec"], frag_tol)
        return dict(cam=cam, tables=tables, pre=pre, binned=binned, fwd=fwd,
                    inputs=tuple(map(self._r, (means, scales, rots, opac))),
                    radii=pre["radii"], allmap=fwd["allmap"])

    def backward(self, st, dL_dallmap, threads=1, want_abs=True):
        grec, gabs = self.render_bwd(st["cam"], st["tables"], st["binned"], st["pre"]["rec"], st["fwd"],
                                     dL_dallmap, want_abs=want_abs, threads=threads)
        means, scales, rots, _ = st["inputs"]
        dm, ds, dr, do = self.preprocess_bwd(st["cam"], means, scales, rots, st["radii"], grec)
        return dict(dmeans=dm, dscales=ds, drots=dr, dopac=do, grec=grec, gabs=gabs)

    # ------------------------------------------------------------------
    def adam(self, p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-15):
        """In-place on p, m, v (contiguous arrays of self.dtype)."""
        assert p.dtype == self.dtype and p.flags.c_contiguous
        self.lib.or_adam(C.c_int64(p.size), _ptr(p), _ptr(self._r(g)), _ptr(m), _ptr(v),
                         C.c_double(lr), C.c_double(b1), C.c_double(b2), C.c_double(eps), C.c_int64(step))

    def atan2(self, y, x):
        y, x = self._r(y), self._r(x)
        out = np.empty_like(y)
        self.lib.or_atan2(C.c_int(y.size), _ptr(y), _ptr(x), _ptr(out))
        return out

    def knn_dist2(self, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        out = np.empty(pts.shape[0], np.float32)
        self.lib.or_knn_dist2(C.c_int(pts.shape[0]), _ptr(pts), _ptr(out))
        return out
