// raster_torch_ext.cpp -- the rasterizer's autograd operator as a compiled torch extension over the C-ABI (include/gvd_raster.h).
//
// What it replaces: the host half of the reference's own torch extension -- `RasterizeGaussiansCUDA` / `RasterizeGaussiansBackwardCUDA`
// (rasterize_points.cu:35-208: output and scratch allocation, argument plumbing) -- plus the Python autograd function on top of it
// (`_RasterizeGaussians`, diff_gaussian_rasterization/__init__.py:44-159), as ONE torch::autograd::Function.  No kernel lives here: the
// compute is libgvd_raster.so, reached through the same C entry points the ctypes binding (_C.py) uses, resolved with dlsym from the
// library path Python hands to init() (so GVD_RASTER_LIB A/B builds and the ctypes side share one library instance and its state).
//
// Why it exists (round 5): with the kernels at ~250 us per training iteration the Python around them -- two ctypes calls with ~40
// marshalled arguments, three ctypes allocator callbacks, ~26 tensor checks, the autograd.Function bookkeeping and the backward's trip
// through the GIL on autograd's device thread -- was 240-300 us per iteration by itself: the loop was host-bound on every box with a
// slower CPU.  Here the backward never touches Python and the forward is one pybind call.
//
// Plain C++ (g++): the only device-side notion is the current stream, taken from c10's HIP stream registry.
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>

#include "../../include/gvd_raster.h"

namespace {

struct Api {
    decltype(&gvd_raster_forward) forward = nullptr;
    decltype(&gvd_raster_backward_conf) backward_conf = nullptr;
    decltype(&gvd_raster_expect_backward) expect_backward = nullptr;
    decltype(&gvd_last_error) last_error = nullptr;
} g_api;

void init(const std::string& path)
{
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    TORCH_CHECK(h != nullptr, "cannot load ", path, ": ", dlerror());
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        TORCH_CHECK(p != nullptr, path, " does not export ", name);
        return p;
    };
    g_api.forward = reinterpret_cast<decltype(g_api.forward)>(sym("gvd_raster_forward"));
    g_api.backward_conf = reinterpret_cast<decltype(g_api.backward_conf)>(sym("gvd_raster_backward_conf"));
    g_api.expect_backward = reinterpret_cast<decltype(g_api.expect_backward)>(sym("gvd_raster_expect_backward"));
    g_api.last_error = reinterpret_cast<decltype(g_api.last_error)>(sym("gvd_last_error"));
}

[[noreturn]] void raise_native(int code)
{
    const char* msg = g_api.last_error ? g_api.last_error() : nullptr;
    TORCH_CHECK(false, "gvd_raster error ", code, ": ", msg ? msg : "?");
}

// float32 contiguous tensor on `dev`, or nullptr for an undefined / empty ("absent") tensor (_C.py: _dev_f32).  `keep` owns a packed copy
// when the caller's tensor is strided.
const float* f32(const at::Tensor& t, const char* name, const c10::Device& dev, at::Tensor& keep)
{
    if (!t.defined() || t.numel() == 0) return nullptr;
    TORCH_CHECK(t.device() == dev, name, " is on ", t.device(), ", expected ", dev, " (no CPU path in this build)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32, got ", t.scalar_type());
    keep = t.is_contiguous() ? t : t.contiguous();
    return keep.data_ptr<float>();
}

struct Chunk {   // allocator callback target: a uint8 tensor sized on demand (resizeFunctional, rasterize_points.cu:27-33)
    at::TensorOptions opt;
    at::Tensor t;
};
char* chunk_alloc(void* user, size_t bytes)
{
    Chunk* c = static_cast<Chunk*>(user);
    c->t = at::empty({(int64_t)bytes}, c->opt);
    return reinterpret_cast<char*>(c->t.data_ptr());
}

struct FwdOut {
    int num_rendered;
    at::Tensor color, depth, alpha, radii, geom, binning, img;
};

FwdOut run_forward(const at::Tensor& bg, const at::Tensor& means3D, const at::Tensor& colors, const at::Tensor& opacity,
                   const at::Tensor& scales, const at::Tensor& rotations, double scale_modifier, const at::Tensor& cov3D,
                   const at::Tensor& viewmatrix, const at::Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t H, int64_t W,
                   const at::Tensor& sh, int64_t degree, const at::Tensor& campos, bool prefiltered, bool debug, bool expect_backward)
{
    TORCH_CHECK(g_api.forward != nullptr, "raster torch extension: init(library path) has not been called");
    TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
    const c10::Device dev = means3D.device();
    TORCH_CHECK(dev.is_cuda(), "diff_gaussian_rasterization (MI355X build) needs tensors on a ROCm device; got ", dev);
    g_api.expect_backward(expect_backward ? 1 : 0);   // per host thread and sticky on the native side; set on every call (no mirror of it
                                                      // here: the ctypes path sets the same flag)
    c10::DeviceGuard guard(dev);
    const int P = (int)means3D.size(0);
    at::Tensor k[11];
    const float *p_bg = f32(bg, "bg", dev, k[0]), *p_m3 = f32(means3D, "means3D", dev, k[1]), *p_col = f32(colors, "colors_precomp", dev, k[2]),
                *p_op = f32(opacity, "opacities", dev, k[3]), *p_sc = f32(scales, "scales", dev, k[4]), *p_rot = f32(rotations, "rotations", dev, k[5]),
                *p_cov = f32(cov3D, "cov3D_precomp", dev, k[6]), *p_vm = f32(viewmatrix, "viewmatrix", dev, k[7]),
                *p_pm = f32(projmatrix, "projmatrix", dev, k[8]), *p_sh = f32(sh, "sh", dev, k[9]), *p_cam = f32(campos, "campos", dev, k[10]);
    const int M = p_sh ? (int)k[9].size(1) : 0;
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    FwdOut o;
    o.color = at::empty({3, H, W}, fopt);
    o.depth = at::empty({1, H, W}, fopt);
    o.alpha = at::empty({1, H, W}, fopt);
    o.radii = at::empty({P}, fopt.dtype(at::kInt));
    Chunk geom{fopt.dtype(at::kByte), at::empty({0}, fopt.dtype(at::kByte))}, binning = geom, img = geom;
    void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    const int rc = g_api.forward(chunk_alloc, &geom, chunk_alloc, &binning, chunk_alloc, &img, P, (int)degree, M, p_bg, (int)W, (int)H, p_m3,
                                 p_sh, p_col, p_op, p_sc, (float)scale_modifier, p_rot, p_cov, p_vm, p_pm, p_cam, (float)tan_fovx,
                                 (float)tan_fovy, prefiltered ? 1 : 0, o.color.data_ptr<float>(), o.depth.data_ptr<float>(),
                                 o.alpha.data_ptr<float>(), P > 0 ? o.radii.data_ptr<int>() : nullptr, debug ? 1 : 0, stream);
    if (rc < 0) raise_native(rc);
    o.num_rendered = rc;
    o.geom = geom.t; o.binning = binning.t; o.img = img.t;
    return o;
}

struct BwdOut {
    at::Tensor means2D, colors, opacity, means3D, cov3D, sh, scales, rotations, conic, depths;
};

// The ten gradient arrays are carved from ONE allocation (64-float aligned starts); the kernels write every element.
BwdOut run_backward(const at::Tensor& bg, const at::Tensor& means3D, const at::Tensor& radii, const at::Tensor& colors, const at::Tensor& scales,
                    const at::Tensor& rotations, double scale_modifier, const at::Tensor& cov3D, const at::Tensor& viewmatrix,
                    const at::Tensor& projmatrix, double tan_fovx, double tan_fovy, const at::Tensor& dL_dcolor, const at::Tensor& dL_ddepth,
                    const at::Tensor& dL_dalpha, const at::Tensor& sh, int64_t degree, const at::Tensor& campos, const at::Tensor& geom,
                    int64_t R, const at::Tensor& binning, const at::Tensor& img, const at::Tensor& alphas, bool debug,
                    const at::Tensor& confidence)
{
    TORCH_CHECK(g_api.backward_conf != nullptr, "raster torch extension: init(library path) has not been called");
    const c10::Device dev = means3D.device();
    TORCH_CHECK(dev.is_cuda(), "diff_gaussian_rasterization (MI355X build) needs tensors on a ROCm device; got ", dev);
    c10::DeviceGuard guard(dev);
    const int64_t P = means3D.size(0);
    TORCH_CHECK(dL_dcolor.defined() && dL_dcolor.dim() == 3, "dL_dout_color must be [3, H, W]");
    const int H = (int)dL_dcolor.size(1), W = (int)dL_dcolor.size(2);
    at::Tensor k[15];
    const float *p_bg = f32(bg, "bg", dev, k[0]), *p_m3 = f32(means3D, "means3D", dev, k[1]), *p_col = f32(colors, "colors_precomp", dev, k[2]),
                *p_sc = f32(scales, "scales", dev, k[3]), *p_rot = f32(rotations, "rotations", dev, k[4]), *p_cov = f32(cov3D, "cov3D_precomp", dev, k[5]),
                *p_vm = f32(viewmatrix, "viewmatrix", dev, k[6]), *p_pm = f32(projmatrix, "projmatrix", dev, k[7]), *p_sh = f32(sh, "sh", dev, k[8]),
                *p_cam = f32(campos, "campos", dev, k[9]), *p_gc = f32(dL_dcolor, "dL_dout_color", dev, k[10]),
                *p_gd = f32(dL_ddepth, "dL_dout_depth", dev, k[11]), *p_ga = f32(dL_dalpha, "dL_dout_alpha", dev, k[12]),
                *p_al = f32(alphas, "alphas", dev, k[13]), *p_conf = f32(confidence, "confidence", dev, k[14]);
    TORCH_CHECK(!p_conf || k[14].numel() == P, "confidence must have ", P, " elements, got ", k[14].sizes());
    const int64_t M = p_sh ? k[8].size(1) : 0;
    const int64_t shapes[10][3] = {{P, 3, 0}, {P, 3, 0}, {P, 3, 0}, {P, 1, 0}, {P, 2, 2}, {P, 1, 0}, {P, 6, 0}, {P, M, 3}, {P, 3, 0}, {P, 4, 0}};
    int64_t off[10], total = 0;
    for (int i = 0; i < 10; i++) {
        const int64_t n = shapes[i][0] * shapes[i][1] * (shapes[i][2] ? shapes[i][2] : 1);
        off[i] = total;
        total += (n + 63) & ~(int64_t)63;
    }
    const at::Tensor flat = at::empty({total > 0 ? total : 1}, at::TensorOptions().dtype(at::kFloat).device(dev));
    auto view = [&](int i) {
        if (shapes[i][2]) return flat.as_strided({shapes[i][0], shapes[i][1], shapes[i][2]}, {shapes[i][1] * shapes[i][2], shapes[i][2], 1}, off[i]);
        return flat.as_strided({shapes[i][0], shapes[i][1]}, {shapes[i][1], 1}, off[i]);
    };
    BwdOut o;
    o.means3D = view(0); o.means2D = view(1); o.colors = view(2); o.depths = view(3); o.conic = view(4); o.opacity = view(5);
    o.cov3D = view(6); o.sh = view(7); o.scales = view(8); o.rotations = view(9);
    if (P != 0) {
        TORCH_CHECK(radii.scalar_type() == at::kInt, "radii must be int32");
        const at::Tensor rad = radii.is_contiguous() ? radii : radii.contiguous();
        void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        const int rc = g_api.backward_conf((int)P, (int)degree, (int)M, (int)R, p_bg, W, H, p_m3, p_sh, p_col, p_al, p_sc, (float)scale_modifier, p_rot,
                                           p_cov, p_vm, p_pm, p_cam, (float)tan_fovx, (float)tan_fovy, rad.data_ptr<int>(),
                                           reinterpret_cast<char*>(geom.data_ptr()), reinterpret_cast<char*>(binning.data_ptr()),
                                           reinterpret_cast<char*>(img.data_ptr()), p_gc, p_gd, p_ga, o.means2D.data_ptr<float>(),
                                           o.conic.data_ptr<float>(), o.opacity.data_ptr<float>(), o.colors.data_ptr<float>(),
                                           o.depths.data_ptr<float>(), o.means3D.data_ptr<float>(), o.cov3D.data_ptr<float>(),
                                           M > 0 ? o.sh.data_ptr<float>() : nullptr, o.scales.data_ptr<float>(), o.rotations.data_ptr<float>(),
                                           p_conf, (size_t)binning.numel(), debug ? 1 : 0, stream);
        if (rc < 0) raise_native(rc);
    }
    return o;
}

// _RasterizeGaussians (diff_gaussian_rasterization/__init__.py:44-159) with the confidence scaling inside the gather kernel.
// Inputs 0-7 are the differentiable ones, in the reference's order; the rest is the settings tuple, flattened.
struct RasterFn : public torch::autograd::Function<RasterFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& means3D, const at::Tensor& means2D,
                                                  const at::Tensor& sh, const at::Tensor& colors, const at::Tensor& opacities,
                                                  const at::Tensor& scales, const at::Tensor& rotations, const at::Tensor& cov3D,
                                                  const at::Tensor& bg, double scale_modifier, const at::Tensor& viewmatrix,
                                                  const at::Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t H, int64_t W,
                                                  int64_t degree, const at::Tensor& campos, bool prefiltered, const at::Tensor& confidence,
                                                  bool expect_backward)
    {
        FwdOut o = run_forward(bg, means3D, colors, opacities, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tan_fovx,
                               tan_fovy, H, W, sh, degree, campos, prefiltered, false, expect_backward);
        ctx->save_for_backward({colors, means3D, scales, rotations, cov3D, o.radii, sh, o.geom, o.binning, o.img, o.alpha, bg, viewmatrix,
                                projmatrix, campos, confidence});
        ctx->saved_data["scale_modifier"] = scale_modifier;
        ctx->saved_data["tan_fovx"] = tan_fovx;
        ctx->saved_data["tan_fovy"] = tan_fovy;
        ctx->saved_data["degree"] = degree;
        ctx->saved_data["num_rendered"] = (int64_t)o.num_rendered;
        ctx->mark_non_differentiable({o.radii});
        ctx->set_materialize_grads(false);   // untouched outputs arrive undefined -> NULL at the C-ABI, no zero fills
        return {o.color, o.radii, o.depth, o.alpha};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads)
    {
        const auto s = ctx->get_saved_variables();
        const at::Tensor &colors = s[0], &means3D = s[1], &scales = s[2], &rotations = s[3], &cov3D = s[4], &radii = s[5], &sh = s[6], &geom = s[7],
                         &binning = s[8], &img = s[9], &alpha = s[10], &bg = s[11], &vm = s[12], &pm = s[13], &campos = s[14], &conf = s[15];
        at::Tensor g_color = grads[0];
        if (!g_color.defined()) g_color = at::zeros({3, alpha.size(1), alpha.size(2)}, alpha.options());
        BwdOut o = run_backward(bg, means3D, radii, colors, scales, rotations, ctx->saved_data["scale_modifier"].toDouble(), cov3D, vm, pm,
                                ctx->saved_data["tan_fovx"].toDouble(), ctx->saved_data["tan_fovy"].toDouble(), g_color, grads[2], grads[3], sh,
                                ctx->saved_data["degree"].toInt(), campos, geom, ctx->saved_data["num_rendered"].toInt(), binning, img, alpha,
                                false, conf);
        const at::Tensor none;
        // (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, then 13 settings entries)
        return {o.means3D, o.means2D, o.sh, o.colors, o.opacity, o.scales, o.rotations, o.cov3D,
                none, none, none, none, none, none, none, none, none, none, none, none, none};
    }
};

std::vector<at::Tensor> rasterize(const at::Tensor& means3D, const at::Tensor& means2D, const at::Tensor& sh, const at::Tensor& colors,
                                  const at::Tensor& opacities, const at::Tensor& scales, const at::Tensor& rotations, const at::Tensor& cov3D,
                                  const at::Tensor& bg, double scale_modifier, const at::Tensor& viewmatrix, const at::Tensor& projmatrix,
                                  double tan_fovx, double tan_fovy, int64_t H, int64_t W, int64_t degree, const at::Tensor& campos,
                                  bool prefiltered, const c10::optional<at::Tensor>& confidence_)
{
    const at::Tensor confidence = confidence_.has_value() ? *confidence_ : at::Tensor();
    // no input wants a gradient (torch.no_grad() / evaluation renders): autograd will never call backward on these buffers, so the
    // forward need not prepare the backward's partial records (gvd_raster.h: gvd_raster_expect_backward)
    bool eb = false;
    if (at::GradMode::is_enabled())
        for (const at::Tensor* t : {&means3D, &means2D, &sh, &colors, &opacities, &scales, &rotations, &cov3D}) eb = eb || (t->defined() && t->requires_grad());
    return RasterFn::apply(means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, bg, scale_modifier, viewmatrix, projmatrix,
                           tan_fovx, tan_fovy, H, W, degree, campos, prefiltered, confidence, eb);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "rasterizer autograd operator over libgvd_raster.so (see csrc/raster_torch_ext.cpp)";
    m.def("init", &init, "resolve the C-ABI entry points from the given library path");
    // (the GIL is released for the call: it waits for the forward's ticket like the ctypes call it replaces, which released it too)
    m.def("rasterize", &rasterize, pybind11::call_guard<pybind11::gil_scoped_release>(),
          "differentiable render: (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, bg, scale_modifier, "
          "viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, sh_degree, campos, prefiltered, confidence) -> "
          "[color, radii, depth, alpha]");
}
