"""Host-side mirror of the reference's torch <-> kernel boundary for this path:

  RasterizeGaussiansCUDA          src/rasterize_points.cu:36-114   (include/rasterize_points.h:18-37)
  RasterizeGaussiansBackwardCUDA  src/rasterize_points.cu:116-193  (include/rasterize_points.h:39-60)
  markVisible                     src/rasterize_points.cu:195-214
  distCUDA2                       third_party/simple-knn/spatial.cu:15-26

Same names, argument order, return tuples and error behaviour; torch supplies device memory
and the current HIP stream only -- all compute happens in libgsr_hip.so behind the C-ABI.
(The LibTorch C++ twin of this file lives in photo-slam_amd/host/.)
"""
import ctypes as C

import torch

from . import capi

# Test-suite hook ONLY: pytest points this at tests/emu/libgsr_emu.so (the same kernel sources compiled
# against the wave64 emulator) so the host logic can be exercised without a GPU.  Anything else is
# refused -- the product path is libgsr_hip.so or an exception, never a CPU implementation.
_LIB_OVERRIDE = None


def _lib():
    if _LIB_OVERRIDE is not None:
        import os
        p = os.path.realpath(_LIB_OVERRIDE)
        if os.path.basename(os.path.dirname(p)) != "emu" or os.path.basename(p) != "libgsr_emu.so" \
                or "PYTEST_CURRENT_TEST" not in os.environ:
            raise RuntimeError("_LIB_OVERRIDE is reserved for the test-suite's emulator build (tests/emu/libgsr_emu.so)")
    return capi.load(_LIB_OVERRIDE)


def _stream_ptr(t):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


def _ptr(t):
    """contiguous().data_ptr() with the reference's empty-tensor -> nullptr convention."""
    if t is None or t.numel() == 0:
        return None, None
    t = t.contiguous()
    return t, C.c_void_p(t.data_ptr())


def _check_device(lib, *tensors):
    backend = lib.gsr_backend()
    for t in tensors:
        if t is None:
            continue
        if backend == b"hip-gfx950" and not t.is_cuda:
            raise RuntimeError("libgsr_hip.so needs device (torch 'cuda' = HIP) tensors; there is no CPU path")
        if backend != b"hip-gfx950" and t.is_cuda:
            raise RuntimeError("the emulator build works on host memory only")


def _bucket_bytes(nbytes):
    """Sizes of a megabyte and more rounded up to the next eighth of the power of two below them (at most 12.5 % more): a training
    run asks for a new binning size at every step, and a caching allocator answers every size it has not seen with a fresh
    hipMalloc; bucketed, the requests recur and are served from its cache (bucket_bytes of the C++ host's rasterize_points.cpp)."""
    nbytes = int(nbytes)
    if nbytes < (1 << 20):
        return nbytes
    step = (1 << (nbytes.bit_length() - 1)) >> 3
    return (nbytes + step - 1) // step * step


def _resize_functional(t):
    """resizeFunctional, src/rasterize_points.cu:28-34 (sizes bucketed: _bucket_bytes)."""
    def fn(_ctx, nbytes):
        t.resize_(_bucket_bytes(nbytes))
        return t.data_ptr()
    return capi.ALLOC_FN(fn)


class RasterWorkspace:
    """Persistent scratch for a caller that renders iteration after iteration (TrainStep; RasterWorkspace of the C++ host's
    rasterize_points.h): the rasterizer's three byte buffers, which the reference allocates per call (src/rasterize_points.cu:71-76).
    They grow with 50 % headroom and never shrink.  The binning buffer follows the instance count, which changes with every step
    of a training run: per-call allocations of ever-new sizes leave the caching allocator with a trail of blocks none of which
    fits the next request (46 GB reserved for 10 GB in use after 300 mapper iterations, against 15 GB; every new largest size is a
    hipMalloc of a gigabyte: up to 45 ms on the pool's boxes)."""

    def __init__(self):
        self.bufs = [None, None, None]   # geometry, binning, image

    def taker(self, i, dev):
        def fn(_ctx, nbytes):
            b = self.bufs[i]
            if b is None or b.device != dev or b.numel() < int(nbytes):
                self.bufs[i] = b = None   # (released first: the two never have to coexist)
                self.bufs[i] = b = torch.empty((int(nbytes) + int(nbytes) // 2,), dtype=torch.uint8, device=dev)
            return b.data_ptr()
        return capi.ALLOC_FN(fn)


def RasterizeGaussiansCUDA(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                           viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                           prefiltered, raw_params=0, sh_adam=None, workspace=None):
    """raw_params (extension, default 0 = reference contract): GSR_RAW_* mask -- opacity / scales / rotations are the
    model's raw parameters and are activated in-kernel (include/gsr.h).
    sh_adam (extension, default None): the dict RasterizeGaussiansBackwardCUDA takes; only its lazy mode (row_step set,
    gsr_sh_adam_lazy) concerns the forward pass: visible rows that lag behind take their missed zero-gradient steps first."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # AT_ERROR, rasterize_points.cu:57-59
    lib = _lib()
    _check_device(lib, means3D, background, viewmatrix, projmatrix, campos)
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    dev = means3D.device
    # torch::full(0) in the reference (rasterize_points.cu:68-69); gsr_forward writes every pixel and every radius itself,
    # so only the P == 0 no-op needs the zeros
    out_color = (torch.empty if P != 0 else torch.zeros)((3, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    geomBuffer = torch.empty((0,), dtype=torch.uint8, device=dev)
    binningBuffer = torch.empty((0,), dtype=torch.uint8, device=dev)
    imgBuffer = torch.empty((0,), dtype=torch.uint8, device=dev)
    rendered = 0
    if P != 0:
        M = sh.size(1) if sh is not None and sh.numel() != 0 else 0
        keep = []
        a = capi.ForwardArgs()
        a.P, a.D, a.M, a.width, a.height = P, int(degree), M, W, H
        a.scale_modifier, a.tan_fovx, a.tan_fovy, a.prefiltered = float(scale_modifier), float(tan_fovx), float(tan_fovy), int(bool(prefiltered))
        a.raw_params = int(raw_params)
        for name, t in (("background", background), ("means3D", means3D), ("shs", sh), ("colors_precomp", colors),
                        ("opacities", opacity), ("scales", scales), ("rotations", rotations),
                        ("cov3D_precomp", cov3D_precomp), ("viewmatrix", viewmatrix), ("projmatrix", projmatrix),
                        ("cam_pos", campos)):
            k, p = _ptr(t.float() if t is not None and t.dtype != torch.float32 else t)
            keep.append(k)
            setattr(a, name, p)
        a.out_color = out_color.data_ptr()
        a.radii = radii.data_ptr()
        if sh_adam is not None and sh_adam.get("row_step") is not None:
            if sh is None or not sh.is_contiguous() or sh.dtype != torch.float32:
                raise RuntimeError("lazy sh_adam needs a contiguous float32 sh tensor (it is updated in place)")
            adam, adam_keep = capi.make_sh_adam(sh, sh_adam)
            keep.append(adam_keep)
            a.sh_adam = C.cast(C.pointer(adam), C.c_void_p)
        if workspace is not None:   # (extension: the caller's persistent buffers, used in place and returned)
            cbs = [workspace.taker(i, dev) for i in range(3)]
        else:
            cbs = [_resize_functional(b) for b in (geomBuffer, binningBuffer, imgBuffer)]
        n = C.c_int(0)
        st = lib.gsr_forward(C.byref(a), cbs[0], None, cbs[1], None, cbs[2], None, _stream_ptr(means3D), C.byref(n))
        capi.check(lib, st, "RasterizeGaussiansCUDA")
        rendered = n.value
        if workspace is not None:
            geomBuffer, binningBuffer, imgBuffer = workspace.bufs
    return rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer


def RasterizeGaussiansBackwardCUDA(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                   viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                   geomBuffer, R, binningBuffer, imageBuffer, raw_params=0, dL_dcolor_view=None, sh_adam=None,
                                   view_stats=None, geom_adam=None, training_outputs_only=False, packed_view=None):
    """dL_dcolor_view (extension, default None = reference contract): a [P,3] float tensor that receives the clamp-masked
    colour gradient; dL_dsh is then NOT computed and None is returned in its place (view-factored gradient exchange,
    shGradFromViews below).
    sh_adam (extension, default None): dict(exp_avg, exp_avg_sq, lr, lr_tail, beta1, beta2, eps, step) -- this step's Adam
    update of `sh` is applied IN PLACE by the kernel that produces its gradient (gsr_backward_args.sh_adam); dL_dsh is then
    not computed and None is returned in its place.
    view_stats (extension, default None): (xyz_gradient_accum, denom, max_radii2D) float tensors with P elements, updated in
    place with this view's densification statistics (gsr_backward_args.stat_*).
    geom_adam (extension, default None): dict(tensors=[(param, exp_avg, exp_avg_sq, lr, step)] for xyz, opacity, scaling,
    rotation, beta1, beta2, eps) -- this step's Adam update of the four geometry tensors is applied IN PLACE by the kernels
    that hold their gradients (gsr_backward_args.geom_adam); dL_dopacity, dL_dmeans3D, dL_dscales and dL_drotations are then
    not computed and None is returned in their places.
    training_outputs_only (extension): dL_dmeans2D and dL_dcov3D are not written either (None returned) -- for a caller that
    fuses the densification statistics (view_stats: the only consumer of dL_dmeans2D in a train step) and optimises scales /
    rotations (no cov3D_precomp)."""
    lib = _lib()
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh is not None and sh.numel() != 0 else 0
    dev = means3D.device
    opts = dict(dtype=torch.float32, device=dev)
    # torch::zeros in the reference (rasterize_points.cu:149-157); gsr_backward writes every element itself
    # the gradients of the four small parameter tensors are slices of ONE buffer (rotation first: its float4 stores need the
    # 16-byte alignment): a data-parallel trainer reduces them over the ranks with a single collective (trainer.py)
    flat = torch.empty((11 * P,), **opts)
    dL_drotations = flat[0:4 * P].view(P, 4)
    dL_dmeans3D = flat[4 * P:7 * P].view(P, 3)
    dL_dscales = flat[7 * P:10 * P].view(P, 3)
    dL_dopacity = flat[10 * P:11 * P].view(P, 1)
    del flat   # (autograd adopts a gradient only if nothing else references it)
    dL_dmeans2D = torch.empty((P, 3), **opts)
    dL_dcolors = torch.empty((P, 3), **opts)
    dL_dconic = torch.empty((P, 2, 2), **opts)
    dL_dcov3D = torch.empty((P, 6), **opts)
    factored = dL_dcolor_view is not None
    if sh_adam is not None:
        if factored and sh_adam.get("row_step") is None:
            # (together only in the lazy form: backward then runs this step's slice of the rows' rotating catch-up, gsr.h)
            raise RuntimeError("sh_adam and dL_dcolor_view are mutually exclusive")
        for t in (sh_adam["exp_avg"], sh_adam["exp_avg_sq"]):
            if t.shape != sh.shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise RuntimeError("sh_adam moments must be contiguous float32 tensors shaped like sh")
        if not sh.is_contiguous():
            raise RuntimeError("sh_adam needs a contiguous sh tensor (it is updated in place)")
    if factored and (dL_dcolor_view.shape != (P, 3) or dL_dcolor_view.dtype != torch.float32 or
                     not dL_dcolor_view.is_contiguous() or dL_dcolor_view.device != dev):
        raise RuntimeError("dL_dcolor_view must be a contiguous float32 (num_points, 3) tensor on the device of means3D")
    dL_dsh = None if factored or sh_adam is not None else torch.empty((P, M, 3), **opts)
    if P != 0:
        keep = []
        a = capi.BackwardArgs()
        a.P, a.D, a.M, a.R, a.width, a.height = P, int(degree), M, int(R), W, H
        a.scale_modifier, a.tan_fovx, a.tan_fovy = float(scale_modifier), float(tan_fovx), float(tan_fovy)
        a.raw_params = int(raw_params)
        for name, t in (("background", background), ("means3D", means3D), ("shs", sh), ("colors_precomp", colors),
                        ("scales", scales), ("rotations", rotations), ("cov3D_precomp", cov3D_precomp),
                        ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("campos", campos),
                        ("radii", radii), ("geom_buffer", geomBuffer), ("binning_buffer", binningBuffer),
                        ("image_buffer", imageBuffer), ("dL_dpix", dL_dout_color)):
            k, p = _ptr(t)
            keep.append(k)
            setattr(a, name, p)
        has_sh = a.shs is not None
        has_scales = a.scales is not None
        a.dL_dmean2D, a.dL_dconic, a.dL_dopacity = dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr()
        a.dL_dcolor, a.dL_dmean3D, a.dL_dcov3D = dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr()
        if factored and not (has_sh and M):
            raise RuntimeError("dL_dcolor_view needs spherical harmonics")
        a.dL_dsh = dL_dsh.data_ptr() if has_sh and M and dL_dsh is not None else None
        if view_stats is not None:
            for t in view_stats:
                if t.numel() != P or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                    raise RuntimeError("view_stats tensors must be contiguous float32 with num_points elements")
            a.stat_grad_accum, a.stat_denom, a.stat_max_radii = (t.data_ptr() for t in view_stats)
        if sh_adam is not None:
            adam, adam_keep = capi.make_sh_adam(sh, sh_adam)
            a.sh_adam = C.pointer(adam)
        a.dL_dcolor_view = dL_dcolor_view.data_ptr() if factored else None
        if packed_view is not None:
            # (message, capacity_rows): a message packViewPlan() prepared -- backward writes its rows and header (gsr_backward_args.packed_view)
            msg, cap = packed_view
            if not factored:
                raise RuntimeError("packed_view needs dL_dcolor_view")
            if msg.dtype != torch.int32 or not msg.is_contiguous() or msg.device != dev or msg.numel() < packedViewWords(P, int(cap)):
                raise RuntimeError("packed_view: a contiguous int32 message of packedViewWords(P, capacity) words on the device of means3D")
            a.packed_view, a.packed_capacity_rows = msg.data_ptr(), int(cap)
        a.dL_dscale = dL_dscales.data_ptr() if has_scales else None
        a.dL_drot = dL_drotations.data_ptr() if has_scales else None
        if training_outputs_only:
            if not has_scales:
                raise RuntimeError("training_outputs_only needs scales / rotations (dL_dcov3D is not written)")
            a.dL_dmean2D = a.dL_dconic = a.dL_dcov3D = None
        if geom_adam is not None:
            for p_, m_, v_, _lr, _step in geom_adam["tensors"]:
                for t in (p_, m_, v_):
                    if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or t.shape != p_.shape:
                        raise RuntimeError("geom_adam tensors must be contiguous float32 on the device of means3D")
            ga = capi.make_geom_adam(geom_adam)
            a.geom_adam = C.pointer(ga)
            a.dL_dopacity = a.dL_dscale = a.dL_drot = None
        st = lib.gsr_backward(C.byref(a), _stream_ptr(means3D))
        capi.check(lib, st, "RasterizeGaussiansBackwardCUDA")
        if not has_sh and dL_dsh is not None:
            dL_dsh.zero_()
        if not has_scales:
            dL_dscales.zero_()
            dL_drotations.zero_()
    if training_outputs_only:
        dL_dmeans2D = dL_dcov3D = None
    if geom_adam is not None:
        dL_dopacity = dL_dmeans3D = dL_dscales = dL_drotations = None
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def _views_args(means3D, campos_views, dL_dcolor_views):
    """pointers + strides (in floats) of the gathered views / centres: dim 0 may be strided (both may be slices of one
    gathered [n_views, P + 1, 3] buffer), the inner dimensions must be dense"""
    P, n_views = means3D.size(0), dL_dcolor_views.size(0)
    if dL_dcolor_views.shape != (n_views, P, 3) or campos_views.shape != (n_views, 3):
        raise RuntimeError("dL_dcolor_views must be (n_views, num_points, 3) and campos_views (n_views, 3)")
    if dL_dcolor_views.dtype != torch.float32 or campos_views.dtype != torch.float32:
        raise RuntimeError("dL_dcolor_views and campos_views must be float32")
    if (P and dL_dcolor_views.stride()[1:] != (3, 1)) or campos_views.stride(1) != 1:
        raise RuntimeError("dL_dcolor_views / campos_views: only the view dimension may be strided")
    return (C.c_void_p(campos_views.data_ptr()), int(campos_views.stride(0)) if n_views > 1 else 3,
            C.c_void_p(dL_dcolor_views.data_ptr()), int(dL_dcolor_views.stride(0)) if n_views > 1 else 3 * P)


def shGradFromViews(means3D, campos_views, dL_dcolor_views, degree, M, scale, out=None):
    """gsr_sh_grad_from_views (include/gsr.h): the [P,M,3] SH gradient of a keyframe batch from the gathered
    [n_views,P,3] dL_dcolor_view outputs and the [n_views,3] camera centres; scale = 1/n_views for the batch mean."""
    lib = _lib()
    P, n_views = means3D.size(0), dL_dcolor_views.size(0)
    pc, sc, pv, sv = _views_args(means3D, campos_views, dL_dcolor_views)
    _check_device(lib, means3D, campos_views, dL_dcolor_views)
    if out is None:
        out = torch.empty((P, M, 3), dtype=torch.float32, device=means3D.device)
    if P != 0:
        k1, p1 = _ptr(means3D)
        st = lib.gsr_sh_grad_from_views(P, int(degree), int(M), n_views, p1, pc, sc, pv, sv, float(scale),
                                        C.c_void_p(out.data_ptr()), _stream_ptr(means3D))
        capi.check(lib, st, "shGradFromViews")
    return out


def shAdamFromViews(means3D, campos_views, dL_dcolor_views, degree, scale, sh, sh_adam):
    """gsr_sh_adam_from_views (include/gsr.h): this step's Adam update of the [P,16,3] tensor `sh` IN PLACE with the batch-mean
    gradient rebuilt from the gathered views; sh_adam as in RasterizeGaussiansBackwardCUDA."""
    lib = _lib()
    P, n_views = means3D.size(0), dL_dcolor_views.size(0)
    pc, sc, pv, sv = _views_args(means3D, campos_views, dL_dcolor_views)
    _check_device(lib, means3D, campos_views, dL_dcolor_views, sh)
    for t in (sh, sh_adam["exp_avg"], sh_adam["exp_avg_sq"]):
        if t.dim() != 3 or t.shape != sh.shape or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("sh and its moments must be contiguous float32 (num_points, M, 3) tensors")
    if P != 0:
        k1, p1 = _ptr(means3D)
        # sh_adam["row_step"] (lazy mode, gsr_sh_adam_lazy): rows no view lights are left alone and step later; the caller runs
        # shAdamLazySlice() after the last row range of the step
        adam, adam_keep = capi.make_sh_adam(sh, sh_adam)
        st = lib.gsr_sh_adam_from_views(P, int(degree), int(sh.size(1)), n_views, p1, pc, sc, pv, sv, float(scale),
                                        C.c_void_p(sh.data_ptr()), C.byref(adam), _stream_ptr(means3D))
        capi.check(lib, st, "shAdamFromViews")


def lastVisibleCount():
    """gsr_last_visible_count: Gaussians with radii > 0 in this thread's last RasterizeGaussiansCUDA (sizes the packed exchange)"""
    return int(_lib().gsr_last_visible_count())


def packedViewWords(P, capacity):
    lib = _lib()
    lib.gsr_packed_view_words.restype = C.c_size_t
    return int(lib.gsr_packed_view_words(int(P), int(capacity)))


def packViewPlan(radii, capacity, message=None):
    """gsr_pack_view_plan (include/gsr.h): the mask and prefix sections of a view's message from the forward pass's radii; the
    backward pass then writes rows and header itself (RasterizeGaussiansBackwardCUDA(packed_view=(message, capacity)))."""
    lib = _lib()
    P = radii.size(0)
    _check_device(lib, radii)
    words = packedViewWords(P, capacity)
    if message is None:
        message = torch.empty(words, dtype=torch.int32, device=radii.device)
    if message.numel() < words or message.dtype != torch.int32 or not message.is_contiguous() or radii.dtype != torch.int32:
        raise RuntimeError("message must be a contiguous int32 tensor of packedViewWords(P, capacity) words, radii int32")
    if P != 0:
        lib.gsr_pack_scratch_bytes.restype = C.c_size_t
        scratch = torch.empty(int(lib.gsr_pack_scratch_bytes(int(P))), dtype=torch.uint8, device=radii.device)
        r = radii.contiguous()
        capi.check(lib, lib.gsr_pack_view_plan(int(P), C.c_void_p(r.data_ptr()), C.c_void_p(message.data_ptr()),
                                               C.c_void_p(scratch.data_ptr()), _stream_ptr(r)), "packViewPlan")
    return message


def packColorView(dL_dcolor_view, campos, capacity, message=None):
    """gsr_pack_color_view (include/gsr.h): the [P,3] colour gradient of one view as a message of its SEEN rows (int32 tensor of
    packedViewWords(P, capacity) words; capacity: a multiple of 4, the same on every rank)."""
    lib = _lib()
    P = dL_dcolor_view.size(0)
    _check_device(lib, dL_dcolor_view, campos)
    words = packedViewWords(P, capacity)
    if message is None:
        message = torch.empty(words, dtype=torch.int32, device=dL_dcolor_view.device)
    if message.numel() < words or message.dtype != torch.int32 or not message.is_contiguous():
        raise RuntimeError("message must be a contiguous int32 tensor of packedViewWords(P, capacity) words")
    if P != 0:
        lib.gsr_pack_scratch_bytes.restype = C.c_size_t
        scratch = torch.empty(int(lib.gsr_pack_scratch_bytes(int(P))), dtype=torch.uint8, device=dL_dcolor_view.device)
        v, c = dL_dcolor_view.contiguous(), campos.contiguous().float()
        capi.check(lib, lib.gsr_pack_color_view(int(P), C.c_void_p(v.data_ptr()), C.c_void_p(c.data_ptr()), int(capacity),
                                                C.c_void_p(message.data_ptr()), C.c_void_p(scratch.data_ptr()), _stream_ptr(v)),
                   "packColorView")
    return message


def checkPackedViews(messages, msg_stride, n_views, P, capacity):
    """gsr_check_packed_views: raises unless every gathered message says "P rows, this capacity, nothing dropped" (waits for the
    current stream: tests, first steps, debugging runs)"""
    lib = _lib()
    _check_device(lib, messages)
    if messages.dtype != torch.int32 or not messages.is_contiguous() or messages.numel() < (int(n_views) - 1) * int(msg_stride) + 8:
        raise RuntimeError("messages must be a contiguous int32 tensor of n_views messages, msg_stride words apart")
    lib.gsr_check_packed_views.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
    st = lib.gsr_check_packed_views(int(P), int(n_views), C.c_void_p(messages.data_ptr()), int(msg_stride), int(capacity),
                                    _stream_ptr(messages))
    if st == -1:   # GSR_ERR_INVALID_ARG
        raise RuntimeError(f"checkPackedViews: a gathered message does not describe {int(P)} rows with capacity {int(capacity)}, "
                           "or its sender dropped rows (include/gsr.h: message word [3])")
    capi.check(lib, st, "checkPackedViews")


def shGradFromPackedViews(means3D, messages, msg_stride, n_views, degree, M, scale):
    """gsr_sh_grad_from_packed_views: shGradFromViews on n_views messages of packColorView, msg_stride words apart"""
    lib = _lib()
    P = means3D.size(0)
    _check_device(lib, means3D, messages)
    out = torch.empty(P, M, 3, dtype=torch.float32, device=means3D.device)
    if P != 0:
        k1, p1 = _ptr(means3D)
        capi.check(lib, lib.gsr_sh_grad_from_packed_views(int(P), int(degree), int(M), int(n_views), p1, C.c_void_p(messages.data_ptr()),
                                                          C.c_longlong(int(msg_stride)), C.c_float(float(scale)),
                                                          C.c_void_p(out.data_ptr()), _stream_ptr(means3D)), "shGradFromPackedViews")
    return out


def shAdamFromPackedViews(means3D, messages, msg_stride, n_views, degree, scale, sh, sh_adam):
    """gsr_sh_adam_from_packed_views: shAdamFromViews on n_views messages of packColorView"""
    lib = _lib()
    P = means3D.size(0)
    _check_device(lib, means3D, messages, sh)
    if P != 0:
        k1, p1 = _ptr(means3D)
        adam, adam_keep = capi.make_sh_adam(sh, sh_adam)
        capi.check(lib, lib.gsr_sh_adam_from_packed_views(int(P), int(degree), int(sh.size(1)), int(n_views), p1,
                                                          C.c_void_p(messages.data_ptr()), C.c_longlong(int(msg_stride)),
                                                          C.c_float(float(scale)), C.c_void_p(sh.data_ptr()), C.byref(adam),
                                                          _stream_ptr(means3D)), "shAdamFromPackedViews")


def shAdamLazySlice(sh, sh_adam, ahead=False):
    """gsr_sh_adam_lazy_slice (include/gsr.h): this step's slice of the row blocks of the lazily stepped [P,16,3] tensor catches
    up -- after the step's shAdamFromViews calls to sh_adam["step"], or (ahead) before them to step - 1 (what the rasterizer's
    backward does by itself in the view-factored mode)."""
    lib = _lib()
    _check_device(lib, sh, sh_adam["exp_avg"], sh_adam["exp_avg_sq"], sh_adam["row_step"])
    with torch.no_grad():
        adam, adam_keep = capi.make_sh_adam(sh, sh_adam)
        capi.check(lib, lib.gsr_sh_adam_lazy_slice(int(sh.size(0)), C.byref(adam), int(bool(ahead)), _stream_ptr(sh)), "shAdamLazySlice")


def adamStepMulti(tensors, beta1, beta2, eps, grad_scale=1.0):
    """gsr_adam_step_multi (include/gsr.h): one Adam step of several tensors in ONE launch.
    tensors: [(param, grad, exp_avg, exp_avg_sq, lr, step), ...] (contiguous float32, at most 8); grad_scale multiplies every
    gradient as it is read (the 1/N of a batch mean whose all-reduce summed)."""
    lib = _lib()
    if not tensors:
        return
    arr = (capi.AdamMultiTensor * len(tensors))()
    for k, (p, g, m, v, lr, step) in enumerate(tensors):
        for t in (p, g, m, v):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel():
                raise RuntimeError("adamStepMulti needs contiguous float32 tensors of one size per entry")
        arr[k] = capi.AdamMultiTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), int(step),
                                      float(grad_scale))
    _check_device(lib, *[t for e in tensors for t in e[:4]])
    capi.check(lib, lib.gsr_adam_step_multi(len(tensors), arr, float(beta1), float(beta2), float(eps), _stream_ptr(tensors[0][0])),
               "adamStepMulti")


def shAdamFlush(sh, sh_adam):
    """gsr_sh_adam_flush (include/gsr.h): every row of `sh` takes the zero-gradient Adam steps it is behind, up to and including
    sh_adam["step"] = the number of steps the tensor has taken (lr / lr_tail belong to that step)."""
    lib = _lib()
    _check_device(lib, sh, sh_adam["exp_avg"], sh_adam["exp_avg_sq"], sh_adam["row_step"])
    if not sh.is_contiguous() or sh.dtype != torch.float32 or sh.dim() != 3 or sh.size(1) != 16:
        raise RuntimeError("shAdamFlush needs a contiguous float32 [P,16,3] tensor")
    with torch.no_grad():
        adam, adam_keep = capi.make_sh_adam(sh, sh_adam)
        st = lib.gsr_sh_adam_flush(int(sh.size(0)), C.byref(adam), _stream_ptr(sh))
        capi.check(lib, st, "shAdamFlush")


def markVisible(means3D, viewmatrix, projmatrix):
    lib = _lib()
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        k1, p1 = _ptr(means3D)
        k2, p2 = _ptr(viewmatrix)
        k3, p3 = _ptr(projmatrix)
        st = lib.gsr_mark_visible(P, p1, p2, p3, C.c_void_p(present.data_ptr()), _stream_ptr(means3D))
        capi.check(lib, st, "markVisible")
    return present


def distCUDA2(points):
    lib = _lib()
    P = points.size(0)
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)
    if P != 0:
        scratch = torch.empty((0,), dtype=torch.uint8, device=points.device)
        cb = _resize_functional(scratch)
        k, p = _ptr(points)
        st = lib.gsr_knn_mean_dist2(P, p, C.c_void_p(means.data_ptr()), cb, None, _stream_ptr(points))
        capi.check(lib, st, "distCUDA2")
    return means
