// hsr_torch_ext.cpp — the torch glue above the C ABI, as a compiled extension (diff_gaussian_rasterization._hsr_torch).
//
// Mirrors the reference's rasterize_points.cu (hierslam-diff-gaussian-rasterization-w-depth/rasterize_points.cu:36-432):
// same four entry points, argument order and return tuples as its pybind module (ext.cpp:15-23), tensors allocated here,
// the three state buffers grown through callbacks (`resizeFunctional`, :27-33), errors as exceptions.  All device work is
// behind include/hsr_rasterizer.h (libhsr_rast.so); this file contains no device code and no HIP calls — the launch
// stream is handed in by the Python wrapper (torch.cuda.current_stream()) and the device is selected with c10's generic
// DeviceGuard.  diff_gaussian_rasterization/_C.py holds the same logic on ctypes and is used when this module has not been
// built; with a 0.65 ms render the ~0.15 ms of interpreter work per call that this removes decides whether the host keeps
// the device fed (bench.py "host").
#include <torch/extension.h>

#include <c10/core/DeviceGuard.h>

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>

#include "../../include/hsr_rasterizer.h"

namespace {

constexpr int NUM_CHANNELS = 3;

[[noreturn]] void fail(int rc, const char* what)
{
    throw std::runtime_error(std::string(what) + " failed (code " + std::to_string(rc) + "): " + hsr_last_error());
}

// grow callback: the buffer is a uint8 tensor owned by the caller's frame (reference resizeFunctional)
char* grow_tensor(size_t bytes, void* user)
{
    try {
        at::Tensor* t = static_cast<at::Tensor*>(user);
        *t = at::empty({(int64_t)bytes}, t->options());
        return reinterpret_cast<char*>(t->data_ptr());
    } catch (...) {
        return nullptr;
    }
}

hsr_buffer as_buffer(at::Tensor& t)
{
    hsr_buffer b;
    b.ptr = t.numel() ? reinterpret_cast<char*>(t.data_ptr()) : nullptr;
    b.capacity = (size_t)t.numel();
    b.grow = grow_tensor;
    b.user = &t;
    return b;
}

// contiguous fp32 tensor on `dev`, or an undefined tensor for the reference's empty placeholders
at::Tensor prep(const c10::optional<at::Tensor>& t, const c10::Device& dev, at::ScalarType dtype = at::kFloat)
{
    if (!t.has_value() || !t->defined() || t->numel() == 0) return at::Tensor();
    TORCH_CHECK(t->device() == dev, "diff_gaussian_rasterization: tensor on ", t->device(), ", expected ", dev);
    TORCH_CHECK(t->scalar_type() == dtype, "diff_gaussian_rasterization: tensor dtype ", t->scalar_type(), ", expected ", dtype);
    return t->contiguous();
}
template <typename T = float>
T* ptr(const at::Tensor& t)
{
    return t.defined() && t.numel() ? reinterpret_cast<T*>(t.data_ptr()) : nullptr;
}

// last num_rendered seen per (device, P, W, H): sizes the binning buffer up front so that the steady state needs no grow
// callback and the forward can enqueue its tail before the read-back returns
std::mutex g_hint_mutex;
std::map<std::tuple<int, int64_t, int64_t, int64_t>, int64_t> g_binning_hint;

using OptT = c10::optional<at::Tensor>;

// -> (rendered, color, aux (semantic map | mask), depth, median_depth, opacity, radii, geomBuffer, binningBuffer, imgBuffer, ticket)
// run_ahead: non-blocking forward (include/hsr_rasterizer.h hsr_forward_arm_async): when the call ran ahead, rendered == HSR_PENDING and
// `ticket` holds the bytes of its hsr_ticket (resolved with forward_end); otherwise `ticket` is empty.
std::tuple<int64_t, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, py::bytes>
forward_common(bool semantic, const OptT& background, const at::Tensor& means3D, const OptT& colors, const OptT& semantics,
               const OptT& opacity, const OptT& scales, const OptT& rotations, double scale_modifier, const OptT& cov3D_precomp,
               const OptT& viewmatrix, const OptT& projmatrix, double tan_fovx, double tan_fovy, int64_t image_height,
               int64_t image_width, const OptT& sh, int64_t degree, const OptT& campos, bool prefiltered, bool debug, int64_t stream,
               bool run_ahead)
{
    TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");   // rasterize_points.cu:60-62
    TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization: tensors must live on a HIP device (got ", means3D.device(),
                "); this build has no CPU path");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0), H = image_height, W = image_width;
    int64_t K = 0;
    if (semantic && semantics.has_value() && semantics->defined() && (semantics->numel() > 0 || semantics->dim() == 2)) {
        // the reference never checks this shape against its compile-time NUM_SEMANTIC (silent OOB)
        TORCH_CHECK(semantics->dim() == 2 && semantics->size(0) == P, "semantics_precomp must have dimensions (num_points, K)");
        K = semantics->size(1);
    }
    c10::DeviceGuard guard(dev);
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    const auto bopt = at::TensorOptions().dtype(at::kByte).device(dev);
    // every pixel of every output is written by the render kernel, so no zero-fill (the reference zero-fills with
    // torch::full first, rasterize_points.cu:71-76)
    at::Tensor out_color = at::empty({NUM_CHANNELS, H, W}, fopt), out_depth = at::empty({1, H, W}, fopt);
    at::Tensor out_median = at::empty({1, H, W}, fopt), out_opacity = at::empty({1, H, W}, fopt);
    at::Tensor out_aux = at::empty({semantic ? K : 1, H, W}, fopt);
    at::Tensor radii = at::empty({P}, fopt.dtype(at::kInt));
    at::Tensor geom, binning, img;
    const auto key = std::make_tuple((int)dev.index(), P, W, H);
    if (P == 0) {
        geom = at::empty({0}, bopt); binning = at::empty({0}, bopt); img = at::empty({0}, bopt);
    } else {
        int64_t hint = 4 * P;
        {
            std::lock_guard<std::mutex> lock(g_hint_mutex);
            auto it = g_binning_hint.find(key);
            if (it != g_binning_hint.end()) hint = it->second;
        }
        geom = at::empty({(int64_t)hsr_required_geometry_bytes((int)P)}, bopt);
        img = at::empty({(int64_t)hsr_required_image_bytes((int)W, (int)H)}, bopt);
        // a call that runs ahead cannot grow the buffer afterwards: twice the last count instead of a quarter more
        binning = at::empty({(int64_t)hsr_required_binning_bytes((int)(run_ahead ? 2 * hint : hint + hint / 4) + 1024)}, bopt);
    }
    const int64_t M = (sh.has_value() && sh->defined() && sh->numel() != 0) ? sh->size(1) : 0;
    const at::Tensor bg_ = prep(background, dev), m3_ = prep(means3D, dev), sh_ = prep(sh, dev), col_ = prep(colors, dev);
    const at::Tensor sem_ = semantic ? prep(semantics, dev) : at::Tensor();
    const at::Tensor op_ = prep(opacity, dev), sc_ = prep(scales, dev), rot_ = prep(rotations, dev), cov_ = prep(cov3D_precomp, dev);
    const at::Tensor vm_ = prep(viewmatrix, dev), pm_ = prep(projmatrix, dev), cp_ = prep(campos, dev);
    hsr_buffer gb = as_buffer(geom), bb = as_buffer(binning), ib = as_buffer(img);
    hsr_ticket tk;
    if (run_ahead && P) hsr_forward_arm_async(&tk);
    int rc;
    {
    // the call waits for the device's num_rendered (up to a whole backward's duration when the stream is busy): other Python threads —
    // a second renderer, autograd's own Python-level backward of another graph — run meanwhile (the ctypes glue releases the GIL for
    // every foreign call by itself); nothing below touches a Python object, the grow callbacks allocate through ATen
    py::gil_scoped_release nogil;
    if (semantic)
        rc = hsr_forward_semantic(&gb, &bb, &ib, (int)P, (int)degree, (int)M, (int)K, ptr(bg_), (int)W, (int)H, ptr(m3_), ptr(sh_),
                                  ptr(col_), ptr(sem_), ptr(op_), ptr(sc_), (float)scale_modifier, ptr(rot_), ptr(cov_), ptr(vm_),
                                  ptr(pm_), ptr(cp_), (float)tan_fovx, (float)tan_fovy, prefiltered ? 1 : 0, ptr(out_color),
                                  ptr(out_aux), ptr(out_depth), ptr(out_median), ptr(out_opacity), ptr<int>(radii), debug ? 1 : 0,
                                  reinterpret_cast<void*>(stream));
    else
        rc = hsr_forward(&gb, &bb, &ib, (int)P, (int)degree, (int)M, ptr(bg_), (int)W, (int)H, ptr(m3_), ptr(sh_), ptr(col_), ptr(op_),
                         ptr(sc_), (float)scale_modifier, ptr(rot_), ptr(cov_), ptr(vm_), ptr(pm_), ptr(cp_), (float)tan_fovx,
                         (float)tan_fovy, prefiltered ? 1 : 0, ptr(out_color), ptr(out_depth), ptr(out_median), ptr(out_opacity),
                         ptr(out_aux), ptr<int>(radii), debug ? 1 : 0, reinterpret_cast<void*>(stream));
    }
    if (rc == HSR_PENDING)
        return std::make_tuple((int64_t)rc, out_color, out_aux, out_depth, out_median, out_opacity, radii, geom, binning, img,
                               py::bytes(reinterpret_cast<const char*>(&tk), sizeof(tk)));
    if (rc < 0) fail(rc, semantic ? "rasterize_gaussians_semantic" : "rasterize_gaussians");
    if (P) {
        std::lock_guard<std::mutex> lock(g_hint_mutex);
        g_binning_hint[key] = rc;
    }
    return std::make_tuple((int64_t)rc, out_color, out_aux, out_depth, out_median, out_opacity, radii, geom, binning, img, py::bytes());
}

// resolves a ticket of a forward call that ran ahead -> (return code of hsr_forward_end, num_rendered as far as known, error text)
std::tuple<int64_t, int64_t, std::string> forward_end(const py::bytes& ticket, bool block, int64_t stream)
{
    const std::string raw = ticket;
    TORCH_CHECK(raw.size() == sizeof(hsr_ticket), "forward_end: not a ticket");
    hsr_ticket tk;
    memcpy(&tk, raw.data(), sizeof(tk));
    int rc;
    {
        py::gil_scoped_release nogil;   // may wait for the device
        rc = hsr_forward_end(&tk, block ? 1 : 0, reinterpret_cast<void*>(stream));
    }
    return std::make_tuple((int64_t)rc, (int64_t)tk.rendered, std::string(rc < 0 && rc != HSR_PENDING ? hsr_last_error() : ""));
}

// -> (dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>
backward_common(bool semantic, const OptT& background, const at::Tensor& means3D, const at::Tensor& radii, const OptT& colors,
                const OptT& semantics, const OptT& scales, const OptT& rotations, double scale_modifier, const OptT& cov3D_precomp,
                const OptT& viewmatrix, const OptT& projmatrix, double tan_fovx, double tan_fovy, const at::Tensor& dL_dout_color,
                const OptT& dL_dout_semantic, const at::Tensor& dL_dout_depth, const at::Tensor& dL_dout_median_depth,
                const at::Tensor& dL_dout_final_opacity, const OptT& sh, int64_t degree, const OptT& campos,
                const at::Tensor& geomBuffer, int64_t R, const at::Tensor& binningBuffer, const at::Tensor& imageBuffer, bool debug,
                bool want_cov3D_grad, bool geometry_only, int64_t stream, const c10::optional<std::vector<OptT>>& sunk)
{
    TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization: tensors must live on a HIP device (got ", means3D.device(),
                "); this build has no CPU path");
    const c10::Device dev = means3D.device();
    const int64_t P = means3D.size(0), H = dL_dout_color.size(1), W = dL_dout_color.size(2);
    const int64_t M = (sh.has_value() && sh->defined() && sh->numel() != 0) ? sh->size(1) : 0;
    const int64_t K = (semantic && dL_dout_semantic.has_value() && dL_dout_semantic->defined()) ? dL_dout_semantic->size(0) : 0;
    c10::DeviceGuard guard(dev);
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    auto fresh = [&](at::IntArrayRef shape) { return P == 0 ? at::zeros(shape, fopt) : at::empty(shape, fopt); };   // fully overwritten when P > 0
    // with a scratch buffer dL_dconic and dL_ddepths are intermediates nobody reads (the reference keeps them inside
    // RasterizeGaussiansBackwardCUDA, rasterize_points.cu:380-383): not allocated, not written
    const size_t nscratch = P ? hsr_backward_scratch_bytes((int)P, (int)K, (int)R) : 0;
    // geometry-only (tracking iteration): no gradient wanted for colours, opacities, semantics, scales, rotations, SH, cov3D
    const bool geo = geometry_only && nscratch && hsr_get_backward_mode() == 0 && colors.has_value() && colors->defined() && colors->numel() != 0;
    // gradient sink (diff_gaussian_rasterization/_C.py set_gradient_sink): pre-allocated outputs — views of a communication
    // bucket — for means3D, colours, semantics, opacities, scales, rotations (checked for shape / dtype / device / layout there)
    auto take = [&](size_t i, at::IntArrayRef shape) {
        if (sunk.has_value() && i < sunk->size() && (*sunk)[i].has_value() && (*sunk)[i]->defined()) return *(*sunk)[i];
        return fresh(shape);
    };
    at::Tensor dL_dmeans3D = take(0, {P, 3}), dL_dmeans2D = fresh({P, 3});
    at::Tensor dL_dcolors, dL_dsemantics, dL_dopacity, dL_dsh, dL_dscales, dL_drotations;
    if (!geo) {
        dL_dcolors = take(1, {P, NUM_CHANNELS}); dL_dsemantics = take(2, {P, K}); dL_dopacity = take(3, {P, 1}); dL_dsh = fresh({P, M, 3});
        dL_dscales = take(4, {P, 3}); dL_drotations = take(5, {P, 4});
    }
    at::Tensor dL_dconic, dL_ddepths, dL_dcov3D, scratch;
    if (!nscratch) { dL_dconic = fresh({P, 2, 2}); dL_ddepths = fresh({P, 1}); }
    if ((want_cov3D_grad && !geo) || P == 0) dL_dcov3D = fresh({P, 6});
    if (P != 0) {
        const at::Tensor bg_ = prep(background, dev), m3_ = prep(means3D, dev), sh_ = prep(sh, dev), col_ = prep(colors, dev);
        const at::Tensor sem_ = semantic ? prep(semantics, dev) : at::Tensor();
        const at::Tensor sc_ = prep(scales, dev), rot_ = prep(rotations, dev), cov_ = prep(cov3D_precomp, dev);
        const at::Tensor vm_ = prep(viewmatrix, dev), pm_ = prep(projmatrix, dev), cp_ = prep(campos, dev);
        const at::Tensor gcol = prep(dL_dout_color, dev), gsem = semantic ? prep(dL_dout_semantic, dev) : at::Tensor();
        const at::Tensor gdep = prep(dL_dout_depth, dev), gmed = prep(dL_dout_median_depth, dev), gop = prep(dL_dout_final_opacity, dev);
        const at::Tensor radii_ = prep(radii, dev, at::kInt);
        if (nscratch) scratch = at::empty({(int64_t)nscratch}, fopt.dtype(at::kByte));
        int rc;
        {
        py::gil_scoped_release nogil;
        if (semantic)
            rc = hsr_backward_semantic((int)P, (int)degree, (int)M, (int)K, (int)R, ptr(bg_), (int)W, (int)H, ptr(m3_), ptr(sh_), ptr(col_),
                                       ptr(sem_), ptr(sc_), (float)scale_modifier, ptr(rot_), ptr(cov_), ptr(vm_), ptr(pm_), ptr(cp_),
                                       (float)tan_fovx, (float)tan_fovy, ptr<int>(radii_), ptr<char>(geomBuffer), ptr<char>(binningBuffer),
                                       ptr<char>(imageBuffer), ptr(gcol), ptr(gsem), ptr(gdep), ptr(gmed), ptr(gop), ptr(dL_dmeans2D),
                                       ptr(dL_dconic), ptr(dL_dopacity), ptr(dL_dcolors), ptr(dL_dsemantics), ptr(dL_ddepths),
                                       ptr(dL_dmeans3D), ptr(dL_dcov3D), ptr(dL_dsh), ptr(dL_dscales), ptr(dL_drotations),
                                       ptr<char>(scratch), nscratch, debug ? 1 : 0, reinterpret_cast<void*>(stream));
        else
            rc = hsr_backward((int)P, (int)degree, (int)M, (int)R, ptr(bg_), (int)W, (int)H, ptr(m3_), ptr(sh_), ptr(col_), ptr(sc_),
                              (float)scale_modifier, ptr(rot_), ptr(cov_), ptr(vm_), ptr(pm_), ptr(cp_), (float)tan_fovx, (float)tan_fovy,
                              ptr<int>(radii_), ptr<char>(geomBuffer), ptr<char>(binningBuffer), ptr<char>(imageBuffer), ptr(gcol),
                              ptr(gdep), ptr(gmed), ptr(gop), ptr(dL_dmeans2D), ptr(dL_dconic), ptr(dL_dopacity), ptr(dL_dcolors),
                              ptr(dL_ddepths), ptr(dL_dmeans3D), ptr(dL_dcov3D), ptr(dL_dsh), ptr(dL_dscales), ptr(dL_drotations),
                              ptr<char>(scratch), nscratch, debug ? 1 : 0, reinterpret_cast<void*>(stream));
        }
        if (rc < 0) fail(rc, semantic ? "rasterize_gaussians_backward_semantic" : "rasterize_gaussians_backward");
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

int64_t binning_hint(int64_t device_index, int64_t P, int64_t W, int64_t H, int64_t set_to)
{
    std::lock_guard<std::mutex> lock(g_hint_mutex);
    const auto key = std::make_tuple((int)device_index, P, W, H);
    if (set_to == -2) { g_binning_hint.erase(key); return -1; }
    if (set_to == -3) { g_binning_hint.clear(); return -1; }   // forget every size (tests: each case starts cold)
    if (set_to >= 0) g_binning_hint[key] = set_to;
    auto it = g_binning_hint.find(key);
    return it == g_binning_hint.end() ? -1 : it->second;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("forward_common", &forward_common);
    m.def("forward_end", &forward_end);
    m.def("backward_common", &backward_common);
    m.def("binning_hint", &binning_hint, "read (-1), set (>= 0) or forget (-2) the binning size hint of (device, P, W, H)");
}
