"""ctypes binding of libgvd_raster.so (include/gvd_raster.h) exposing the three entry points of
the reference's pybind module (ext.cpp:15-18) with the same argument order and return tuples
(rasterize_points.h:19-70):

    rasterize_gaussians(...)          -> (num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer)
    rasterize_gaussians_backward(...) -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]

There is NO CPU fallback: tensors must live on a ROCm device and the HIP library must be built
(python __graft_entry__.py build); anything else raises.
"""
import ctypes
import os
import threading

import torch  # must be imported before the .so so that ONE libamdhip64 (torch's) serves both

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GVD_RASTER_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libgvd_raster.so")  # env: A/B builds
_lib = None

_ALLOC = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
_F = ctypes.c_float
_I = ctypes.c_int
_P = ctypes.c_void_p


class _ChunkLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in (
        "depths", "means2D", "conic_opacity", "rgbd", "cov3D", "clamped", "internal_radii", "tiles_touched",
        "point_offsets", "scalars", "ranges", "n_contrib", "point_list_keys", "point_list", "bucket", "tile_order")]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: build the HIP extension first "
                               f"(python -c 'import __graft_entry__ as g; g.build()')")
        L = ctypes.CDLL(_LIB_PATH)
        L.gvd_last_error.restype = ctypes.c_char_p
        L.gvd_version.restype = ctypes.c_char_p
        L.gvd_raster_geometry_bytes.restype = ctypes.c_size_t
        L.gvd_raster_image_bytes.restype = ctypes.c_size_t
        L.gvd_raster_binning_bytes.restype = ctypes.c_size_t
        L.gvd_raster_binning_bytes.argtypes = [ctypes.c_uint32]
        L.gvd_raster_forward.restype = _I
        L.gvd_raster_forward.argtypes = [_ALLOC, _P, _ALLOC, _P, _ALLOC, _P, _I, _I, _I, _P, _I, _I,
                                         _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _F, _F, _I,
                                         _P, _P, _P, _P, _I, _P]
        L.gvd_raster_forward_capped.restype = _I
        L.gvd_raster_forward_capped.argtypes = [_P, _P, _P, ctypes.c_uint32, _I, _I, _I, _P, _I, _I,
                                                _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _F, _F, _I,
                                                _P, _P, _P, _P, _P, _I, _P]
        _bw = [_I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P,
               _F, _F, _P, _P, _P, _P, _P, _P, _P,
               _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]
        L.gvd_raster_backward.restype = _I
        L.gvd_raster_backward.argtypes = _bw + [_I, _P]
        L.gvd_raster_backward_conf.restype = _I
        L.gvd_raster_backward_conf.argtypes = _bw + [_P, ctypes.c_size_t, _I, _P]
        L.gvd_raster_binning_capacity.restype = ctypes.c_uint32
        L.gvd_raster_binning_capacity.argtypes = [ctypes.c_size_t]
        L.gvd_raster_set_speculation.argtypes = [_I]
        L.gvd_raster_expect_backward.argtypes = [_I]
        L.gvd_raster_binning_bytes_no_backward.restype = ctypes.c_size_t
        L.gvd_raster_binning_bytes_no_backward.argtypes = [ctypes.c_uint32]
        L.gvd_raster_set_backward_split.argtypes = [_I]
        # this binding hands the binning chunk's size to backward, so the forward may lay it out speculatively (gvd_raster.h)
        L.gvd_raster_set_speculation(1)
        L.gvd_raster_mark_visible.restype = _I
        L.gvd_raster_mark_visible.argtypes = [_I, _P, _P, _P, _P, _P]
        L.gvd_raster_chunk_layout.argtypes = [_I, _I, _I, ctypes.c_uint32, ctypes.POINTER(_ChunkLayout)]
        L.gvd_profile_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I)]
        _lib = L
    return _lib


_EXT_PATH = os.path.join(os.path.dirname(_HERE), "lib", "_gvd_raster_torch.so")
_ext = None


def ext():
    """The compiled autograd operator (csrc/raster_torch_ext.cpp: forward + backward as one torch::autograd::Function over the same
    C-ABI library instance), or None when lib/_gvd_raster_torch.so has not been built -- the ctypes functions below then carry the
    operator (same kernels, ~2x the host time per training iteration).  GVD_RASTER_NO_EXT=1 forces that path (A/B runs)."""
    global _ext
    if _ext is None:
        if os.environ.get("GVD_RASTER_NO_EXT") or not os.path.exists(_EXT_PATH):
            _ext = False
        else:
            import importlib.util
            lib()   # the HIP library first: a missing library must raise its own message
            spec = importlib.util.spec_from_file_location("_gvd_raster_torch", _EXT_PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.init(_LIB_PATH)
            _ext = mod
    return _ext or None


def _err(code):
    msg = lib().gvd_last_error()
    return RuntimeError(f"gvd_raster error {code}: {msg.decode() if msg else '?'}")


_F32 = torch.float32


def _dev_f32(t, name, device):
    """float32 contiguous tensor on `device`, or None for an empty ('absent') tensor.  (The common case -- right device, float32,
    contiguous -- is three attribute reads; this function runs ~26 times per training iteration, on the path that feeds the GPU.)"""
    if t is None:
        return None
    if t.device == device and t.dtype is _F32 and t.is_contiguous():
        return t if t.numel() else None
    if t.numel() == 0:
        return None
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device} (no CPU path in this build)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw hipStream_t of torch's current stream on the current device (the private fast path costs ~0.3 us, the public
    torch.cuda.current_stream() object ~9 us)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _Noop:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOOP = _Noop()


def _on(dev):
    """Device guard; a no-op when `dev` is already current (torch.cuda.device() costs ~25 us of host time per use,
    which matters in the 150-us window the host has to queue the backward behind the forward)."""
    idx = dev.index
    return _NOOP if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(dev)


class _Chunk:
    """Allocator callback target: a torch uint8 tensor sized on demand (resizeFunctional,
    rasterize_points.cu:27-33)."""

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.cb = _ALLOC(self._alloc)

    def _alloc(self, _user, nbytes):
        self.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.tensor.data_ptr()


# ---- opt-in sync-free forward -----------------------------------------------------------------------------------
# The reference's forward returns num_rendered as a host int (rasterize_points.cu:35-115), which costs one device->host
# round trip per render and stops the host from queueing the loss and the backward while the forward still runs.
# With an instance capacity set (set_instance_capacity(n) or GVD_RASTER_CAPACITY=n) the binning chunk is sized for n
# instances up front, nothing is read back, `num_rendered` is reported as n, and an overflow (more than n instances)
# raises at the NEXT rasterize call of the process (the flag has landed by then; that render showed background only).
_CAPACITY = int(os.environ.get("GVD_RASTER_CAPACITY", "0") or 0)
_PENDING = []   # (event, pinned int32 tensor) of earlier capped calls whose status has not been looked at yet


def set_instance_capacity(n):
    """0 = reference behaviour (one sync per forward, exact num_rendered); n > 0 = sync-free forward for up to n
    (Gaussian, tile) instances per render."""
    global _CAPACITY
    _CAPACITY = int(n)


def _drain_status(block=False):
    while _PENDING:
        ev, host = _PENDING[0]
        if not block and not ev.query():
            return
        ev.synchronize()
        _PENDING.pop(0)
        if int(host[0]) != 0:
            raise RuntimeError(f"diff_gaussian_rasterization: a sync-free forward exceeded the instance capacity "
                               f"({_CAPACITY}); raise it with set_instance_capacity() / GVD_RASTER_CAPACITY")


_EMPTY = torch.empty(0, dtype=torch.uint8)
_TLS = threading.local()


def _chunks(dev):
    """Three allocator-callback targets per (thread, device), created once."""
    cache = getattr(_TLS, "chunks", None)
    if cache is None:
        cache = _TLS.chunks = {}
    c = cache.get(dev)
    if c is None:
        c = cache[dev] = (_Chunk(dev), _Chunk(dev), _Chunk(dev))
    return c


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, expect_backward=True):
    """expect_backward (MI355X addition, keyword only in spirit): False = no rasterize_gaussians_backward will be run on the
    returned buffers (no-grad render) -> the forward skips preparing the backward's partial records (gvd_raster.h)."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("diff_gaussian_rasterization (MI355X build) needs tensors on a ROCm device; got " + str(dev))
    L = lib()
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    L.gvd_raster_expect_backward(1 if expect_backward else 0)   # per host thread, sticky on the native side; set on every call (the
                                                                # compiled operator sets the same flag: no mirror of it is kept here)
    with _on(dev):
        f = lambda t, n: _dev_f32(t, n, dev)
        bg, m3, col, opa, sc, rot, cov, vm, pm, shs, cam = (
            f(background, "bg"), f(means3D, "means3D"), f(colors, "colors_precomp"), f(opacity, "opacities"),
            f(scales, "scales"), f(rotations, "rotations"), f(cov3D_precomp, "cov3D_precomp"), f(viewmatrix, "viewmatrix"),
            f(projmatrix, "projmatrix"), f(sh, "sh"), f(campos, "campos"))
        M = 0 if shs is None else shs.size(1)
        out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        out_alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        if _CAPACITY > 0 and P > 0:
            _drain_status()
            u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
            geom_t, img_t = u8(L.gvd_raster_geometry_bytes(P, W, H)), u8(L.gvd_raster_image_bytes(W, H))
            # (a forward no backward will follow touches keys / list / bucket / cull bytes only: the compact form holds those at the same offsets)
            bin_t = u8(L.gvd_raster_binning_bytes(_CAPACITY) if expect_backward else L.gvd_raster_binning_bytes_no_backward(_CAPACITY))
            status = torch.empty(1, dtype=torch.int32, device=dev)
            rc = L.gvd_raster_forward_capped(geom_t.data_ptr(), bin_t.data_ptr(), img_t.data_ptr(), _CAPACITY, P, int(degree), M,
                                             _ptr(bg), W, H, _ptr(m3), _ptr(shs), _ptr(col), _ptr(opa), _ptr(sc),
                                             float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cam),
                                             float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), out_color.data_ptr(),
                                             out_depth.data_ptr(), out_alpha.data_ptr(), radii.data_ptr(), status.data_ptr(),
                                             int(bool(debug)), _stream())
            if rc < 0:
                raise _err(rc)
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(status, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            _PENDING.append((ev, host))
            return _CAPACITY, out_color, out_depth, out_alpha, radii, geom_t, bin_t, img_t
        geom, binning, img = _chunks(dev)   # allocator callback targets, reused (building a ctypes callback costs ~5 us)
        pre = _grad_arrays(P, M, dev) if (expect_backward and P > 0) else None
        rc = L.gvd_raster_forward(geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), M, _ptr(bg), W, H,
                                  _ptr(m3), _ptr(shs), _ptr(col), _ptr(opa), _ptr(sc), float(scale_modifier), _ptr(rot),
                                  _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cam), float(tan_fovx), float(tan_fovy),
                                  int(bool(prefiltered)), out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr(),
                                  radii.data_ptr() if P > 0 else None, int(bool(debug)), _stream())
        if rc < 0:
            raise _err(rc)
    res = (rc, out_color, out_depth, out_alpha, radii, geom.tensor, binning.tensor, img.tensor)
    if pre is not None and geom.tensor.numel():
        _park_grad_arrays(geom.tensor.data_ptr(), P, M, pre)
    geom.tensor = binning.tensor = img.tensor = _EMPTY   # the chunks now belong to the caller only
    return res


# The backward's ten gradient arrays, carved from ONE allocation, 256-byte aligned starts; the kernels write every element.
_GRAD_LAYOUTS = {}


def _grad_layout(P, M):
    """(total floats, [(shape, stride, offset)] x 10) -- computed once per (P, M)."""
    lay = _GRAD_LAYOUTS.get((P, M))
    if lay is None:
        shapes = ((P, 3), (P, 3), (P, 3), (P, 1), (P, 2, 2), (P, 1), (P, 6), (P, M, 3), (P, 3), (P, 4))
        items, total = [], 0
        for sh in shapes:
            n, stride = 1, []
            for d in reversed(sh):
                stride.append(n)
                n *= d
            items.append((sh, tuple(reversed(stride)), total))
            total += (n + 63) & ~63
        if len(_GRAD_LAYOUTS) > 64:
            _GRAD_LAYOUTS.clear()
        lay = _GRAD_LAYOUTS[(P, M)] = (max(total, 1), tuple(items))
    return lay


def _grad_arrays(P, M, dev):
    total, items = _grad_layout(P, M)
    flat = torch.empty((total,), dtype=torch.float32, device=dev)
    return tuple(flat.as_strided(sh, st, off) for sh, st, off in items)


# A forward that expects a backward allocates the backward's outputs BEFORE its native call -- while the GPU still works on the
# previous iteration -- and parks them under the geometry chunk's address: the backward wrapper runs inside the ~80 us of GPU work the
# forward left queued, and allocator calls there are GPU idle time once that runs out (measured: the wrapper 62 -> ~35 us).  The
# backward runs on autograd's device thread, so the hand-over is a small process-wide dict, not thread-local state.
_PREALLOC = {}
_PREALLOC_LOCK = threading.Lock()


def _park_grad_arrays(key, P, M, arrays):
    with _PREALLOC_LOCK:
        if len(_PREALLOC) >= 8:
            _PREALLOC.pop(next(iter(_PREALLOC)))
        _PREALLOC[key] = (P, M, arrays)


def _take_grad_arrays(key, P, M, dev):
    with _PREALLOC_LOCK:
        hit = _PREALLOC.pop(key, None)
    if hit is not None and hit[0] == P and hit[1] == M and hit[2][0].device == dev:
        return hit[2]
    return _grad_arrays(P, M, dev)


KEEP_BACKWARD_INTERNALS = False
LAST_BACKWARD_INTERNALS = {}


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, dL_dout_alpha,
                                 sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug,
                                 confidence=None):
    """Reference signature (rasterize_points.h:41-65) + optional `confidence` [P,1]: when given, the
    returned gradients (all but dL_dmeans2D) are already multiplied by it inside the gather kernel.
    dL_dout_depth / dL_dout_alpha may be None (zero gradient)."""
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("diff_gaussian_rasterization (MI355X build) needs tensors on a ROCm device; got " + str(dev))
    L = lib()
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    with _on(dev):
        f = lambda t, n: _dev_f32(t, n, dev)
        bg, m3, col, sc, rot, cov, vm, pm, shs, cam = (
            f(background, "bg"), f(means3D, "means3D"), f(colors, "colors_precomp"), f(scales, "scales"),
            f(rotations, "rotations"), f(cov3D_precomp, "cov3D_precomp"), f(viewmatrix, "viewmatrix"),
            f(projmatrix, "projmatrix"), f(sh, "sh"), f(campos, "campos"))
        gC, gD, gA, al = f(dL_dout_color, "dL_dout_color"), f(dL_dout_depth, "dL_dout_depth"), f(dL_dout_alpha, "dL_dout_alpha"), f(alphas, "alphas")
        conf = f(confidence, "confidence")
        if conf is not None and conf.numel() != P:
            raise RuntimeError(f"confidence must have {P} elements, got {tuple(conf.shape)}")
        M = 0 if shs is None else shs.size(1)
        (dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_ddepths, dL_dconic, dL_dopacity, dL_dcov3D, dL_dsh, dL_dscales,
         dL_drotations) = _take_grad_arrays(geomBuffer.data_ptr(), P, M, dev)
        if P != 0:
            rc = L.gvd_raster_backward_conf(P, int(degree), M, int(R), _ptr(bg), W, H, _ptr(m3), _ptr(shs), _ptr(col), _ptr(al),
                                       _ptr(sc), float(scale_modifier), _ptr(rot), _ptr(cov), _ptr(vm), _ptr(pm), _ptr(cam),
                                       float(tan_fovx), float(tan_fovy), radii.contiguous().data_ptr(),
                                       geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr(),
                                       _ptr(gC), _ptr(gD), _ptr(gA), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(),
                                       dL_dopacity.data_ptr(), dL_dcolors.data_ptr(), dL_ddepths.data_ptr(),
                                       dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(), dL_dsh.data_ptr() if M > 0 else None,
                                       dL_dscales.data_ptr(), dL_drotations.data_ptr(), _ptr(conf), int(binningBuffer.numel()),
                                       int(bool(debug)), _stream())
            if rc < 0:
                raise _err(rc)
    if KEEP_BACKWARD_INTERNALS:   # parity tests: the per-Gaussian sums the reference keeps internal (rasterize_points.cu:172-176)
        LAST_BACKWARD_INTERNALS.update(dL_dconic=dL_dconic, dL_ddepths=dL_ddepths)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("diff_gaussian_rasterization (MI355X build) needs tensors on a ROCm device; got " + str(dev))
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        with _on(dev):
            rc = lib().gvd_raster_mark_visible(P, _dev_f32(means3D, "means3D", dev).data_ptr(),
                                               _dev_f32(viewmatrix, "viewmatrix", dev).data_ptr(),
                                               _dev_f32(projmatrix, "projmatrix", dev).data_ptr(),
                                               present.data_ptr(), _stream())
            if rc < 0:
                raise _err(rc)
    return present


def chunk_views(P, W, H, R, geomBuffer, binningBuffer, imgBuffer):
    """Typed views of the internal scratch arrays (tests / debugging only)."""
    lay = _ChunkLayout()
    L = lib()
    cap = L.gvd_raster_binning_capacity(binningBuffer.numel()) if binningBuffer.numel() else int(R)
    if cap == 0xffffffff:
        raise RuntimeError("binningBuffer does not have the size of a binning chunk")
    if cap < int(R):
        raise RuntimeError("binningBuffer is smaller than num_rendered requires")
    L.gvd_raster_chunk_layout(P, W, H, cap, ctypes.byref(lay))

    def view(buf, off, nbytes, dtype):
        base = buf.data_ptr()
        a = (-base) % 128
        return buf[a + off:a + off + nbytes].view(dtype)

    T = ((W + 15) // 16) * ((H + 15) // 16)
    g, b, i = geomBuffer, binningBuffer, imgBuffer
    return dict(
        depths=view(g, lay.depths, 4 * P, torch.float32),
        means2D=view(g, lay.means2D, 8 * P, torch.float32).view(P, 2),
        conic_opacity=view(g, lay.conic_opacity, 16 * P, torch.float32).view(P, 4),
        rgbd=view(g, lay.rgbd, 16 * P, torch.float32).view(P, 4),
        cov3D=view(g, lay.cov3D, 24 * P, torch.float32).view(P, 6),
        clamped=view(g, lay.clamped, 4 * P, torch.int32),
        tiles_touched=view(g, lay.tiles_touched, 4 * P, torch.int32),
        point_offsets=view(g, lay.point_offsets, 4 * P, torch.int32),
        scalars=view(g, lay.scalars, 32, torch.int32),
        ranges=view(i, lay.ranges, 8 * T, torch.int32).view(T, 2),
        n_contrib=view(i, lay.n_contrib, 4 * W * H, torch.int32).view(H, W),
        tile_order=view(i, lay.tile_order, 4 * T, torch.int32),
        point_list_keys=view(b, lay.point_list_keys, 8 * R, torch.int64) if R > 0 else None,
        point_list=view(b, lay.point_list, 4 * R, torch.int32) if R > 0 else None,
    )
